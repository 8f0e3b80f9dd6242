#!/bin/bash
# same-box A/B of two builds of the library: tools/_ab/libgpirt_hip_head.so against the tree's
out=gpurun_out/ab2; mkdir -p $out
cp gpirt_amd/libgpirt_hip.so /tmp/new.so
one() { lib=$1; name=$2; shift 2; cp $lib gpirt_amd/libgpirt_hip.so
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-variants --steps 30 --warmup 5 > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -3 $out/$name.err; return 1; }
  python - "$out/$name.json" "$name" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); p=d["phases_ms"]; r=d["roofline"]
print(sys.argv[2].ljust(16), "%.2f sweeps/s" % d["value"], "potrf %.2f nu %.2f" % (p["potrf"], p["nu"]), "upd %.1f TF" % r["achieved"], "whole-nu", r["other_instances"].get("nu_whole_product_in_one_launch",{}).get("ms_per_launch"), "chol alone", r.get("cholesky_phase_alone",{}).get("ms"))
P
}
for i in 1 2; do
one tools/_ab/libgpirt_hip_head.so head$i GPIRT_X=0 && one /tmp/new.so new$i GPIRT_LOOKAHEAD=0 && one /tmp/new.so new_nl0_$i GPIRT_LOOKAHEAD=0 GPIRT_NU_LONE=0 || break
done
cp /tmp/new.so gpirt_amd/libgpirt_hip.so
