#!/bin/bash
# One bench line plus the rocprofv3 kernel statistics of the same command.
# usage: tools/profile_round.sh OUTDIR   (writes OUTDIR/bench.json, OUTDIR/kernel_stats.csv)
out=$1
mkdir -p $out
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 python -u $R/bench.py --steps 20 --warmup 3 > $R/$out/bench.json 2> $R/$out/bench.err || { tail -5 $R/$out/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- \
  python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants > $R/$out/prof.log 2>&1 || { tail -5 $R/$out/prof.log; exit 1; }
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $R/$out/kernel_stats.csv
head -12 $R/$out/kernel_stats.csv
