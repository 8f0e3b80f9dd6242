#!/bin/bash
mkdir -p gpurun_out/mscale
for M in 1024 2048 4096 8192; do
  timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --n 10000 --m $M --no-cpu-baseline --no-variants > gpurun_out/mscale/m$M.json 2> gpurun_out/mscale/m$M.err || { tail -5 gpurun_out/mscale/m$M.err; exit 1; }
  python - $M gpurun_out/mscale/m$M.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], round(d["value"], 2), {k: round(v, 3) for k, v in d["phases_ms"].items() if v > 0.05}, d["likelihood_evals_per_step"])
PY
done
