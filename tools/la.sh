#!/bin/bash
# A/B of the factorisation's look-ahead schedule: bench lines with and without it, with and without
# the rows of nu beside the factorisation
out=gpurun_out/la; mkdir -p $out
run() { name=$1; shift; env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-variants --steps 30 --warmup 5 > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -5 $out/$name.err; return 1; }
  python - "$out/$name.json" "$name" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); p=d["phases_ms"]
print(sys.argv[2].ljust(28), "%.2f sweeps/s" % d["value"], "potrf %.2f nu %.2f" % (p["potrf"], p["nu"]), "upd %.1f TF" % d["roofline"]["achieved"], "x%.0f" % d["roofline"]["launches_per_step"], "alone", d["roofline"].get("cholesky_phase_alone",{}).get("ms"))
P
}
run base GPIRT_LOOKAHEAD=0 && run ahead GPIRT_LOOKAHEAD=1 && run base_noov GPIRT_LOOKAHEAD=0 GPIRT_NU_OVERLAP=0 && run ahead_noov GPIRT_LOOKAHEAD=1 GPIRT_NU_OVERLAP=0
run ahead_nulone0 GPIRT_LOOKAHEAD=1 GPIRT_NU_LONE=0 && run base_nulone0 GPIRT_LOOKAHEAD=0 GPIRT_NU_LONE=0
