#!/bin/bash
# usage: tools/ab_size.sh OUTDIR N M "ENV1=.." "ENV.." ...  -- one bench run (50 steps) per environment string at (N, M)
out=$1; n=$2; m=$3; shift 3
mkdir -p $out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python -u bench.py --steps 50 --warmup 10 --n $n --m $m --no-cpu-baseline --no-variants > $out/s$i.json 2> $out/s$i.err || { tail -5 $out/s$i.err; exit 1; }
  python - "$e" $out/s$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "|", round(d["value"], 2), "sweeps/s", {k: round(v, 3) for k, v in d["phases_ms"].items() if v > 0.02})
PY
done
