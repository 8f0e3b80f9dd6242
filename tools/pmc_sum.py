"""Mean per-launch value of one counter for the kernels whose name contains argv[3], from a
rocprofv3 --pmc run directory argv[1]; argv[2] = counter name."""
import csv
import glob
import sys

vals = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2] and sys.argv[3] in r["Kernel_Name"]:
            vals.setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
print(sys.argv[2], "launches", len(vals), "mean", sum(vals.values()) / max(1, len(vals)))
