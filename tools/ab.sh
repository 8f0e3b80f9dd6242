#!/bin/bash
# usage: tools/ab.sh OUTDIR "ENV1=.. ENV2=.." "ENV.." ...   -- one bench run (20 steps) per environment string
out=$1; shift
mkdir -p $out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants > $out/ab$i.json 2> $out/ab$i.err || { tail -5 $out/ab$i.err; exit 1; }
  python - "$e" $out/ab$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "|", round(d["value"], 2), "sweeps/s", {k: round(v, 2) for k, v in d["phases_ms"].items() if v > 0.05},
      "rows beside", d["substeps"]["potrf"]["rows_of_nu_computed_beside_it"])
PY
done
