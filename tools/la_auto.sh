#!/bin/bash
# the sampler's own choice of schedule at the full size and at a rank's share of the items
out=gpurun_out/la_auto; mkdir -p $out
for m in 4096 2048 1024 512; do
  name=m${m}
  timeout -k 10 150 python bench.py --m $m --no-cpu-baseline --no-variants --steps 30 --warmup 5 > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -3 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); p=d["phases_ms"]
print(sys.argv[2].ljust(8), "%.2f sweeps/s %.2f ms" % (d["value"], d["ms_per_step"]), "potrf %.2f nu %.2f" % (p["potrf"], p["nu"]), "rows beside", d["substeps"]["potrf"]["rows_of_nu_computed_beside_it"], "lookahead", d["substeps"]["potrf"]["lookahead"])
P
done
