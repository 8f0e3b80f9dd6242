#!/bin/bash
# usage: tools/kstats.sh OUTDIR [ENV=VAL ...] : rocprofv3 kernel statistics of a short bench run
out=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- \
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants $BENCH_ARGS > $R/$out/prof.log 2>&1 || { tail -5 $R/$out/prof.log; exit 1; }
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $R/$out/kernel_stats.csv
python3 - $R/$out/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
