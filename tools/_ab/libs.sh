#!/bin/bash
# same-box A/B of several builds of the library (tools/_ab/*.so), each at the sizes given in SIZES
out=gpurun_out/ab3; mkdir -p $out
cp gpirt_amd/libgpirt_hip.so /tmp/tree.so
for rep in 1 2; do for lib in $LIBS; do for m in ${SIZES:-4096 1024}; do
  name=${lib}_m${m}_$rep; cp tools/_ab/$lib.so gpirt_amd/libgpirt_hip.so
  env $EXTRA timeout -k 10 150 python bench.py --m $m --no-cpu-baseline --no-variants --steps 30 --warmup 5 > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -3 $out/$name.err; cp /tmp/tree.so gpirt_amd/libgpirt_hip.so; exit 1; }
  python - "$out/$name.json" "$name" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); p=d["phases_ms"]; r=d["roofline"]
print(sys.argv[2].ljust(20), "%.2f sweeps/s" % d["value"], "potrf %.2f nu %.2f" % (p["potrf"], p["nu"]), "upd %.1f TF" % r["achieved"], "rows", d["substeps"]["potrf"]["rows_of_nu_computed_beside_it"], "la", d["substeps"]["potrf"]["lookahead"])
P
done; done; done
cp /tmp/tree.so gpirt_amd/libgpirt_hip.so
