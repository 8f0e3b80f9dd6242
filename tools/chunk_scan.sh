#!/bin/bash
# Scan of the chunking of nu = L Z beside the factorisation: tile rows per chunk, super-panels beside,
# tail panels.  usage: tools/chunk_scan.sh "ENV=.. ENV=.." "ENV=.." ...   (one bench line per argument)
out=gpurun_out/chunk; mkdir -p $out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 150 python bench.py $BENCH_ARGS --no-cpu-baseline --no-variants --steps 30 --warmup 5 > $out/$i.json 2> $out/$i.err || { echo "FAILED $cfg"; tail -5 $out/$i.err; exit 1; }
  python - "$out/$i.json" "$cfg" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); p=d["phases_ms"]
print(sys.argv[2].ljust(60), "%.2f sweeps/s" % d["value"], "potrf %.2f nu %.2f" % (p["potrf"], p["nu"]), "rows", d["substeps"]["potrf"]["rows_of_nu_computed_beside_it"], "la", d["substeps"]["potrf"]["lookahead"])
P
done
