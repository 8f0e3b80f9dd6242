"""Print (calls, average us) of the kernels whose name contains argv[2] from a rocprofv3
kernel_stats.csv found under directory argv[1]."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            print(r["Name"][:50], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
