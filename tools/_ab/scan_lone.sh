#!/bin/bash
# scan the lone grid size together with the chunk height of nu (tile rows x 32 column tiles = workgroups)
cd $GRAFT_REPO_ROOT
for cfg in "7 224 1" "6 192 1" "5 160 1" "6 192 0" "7 224 0"; do
  set -- $cfg
  GPIRT_POTRF_STEPWISE=-1 GPIRT_NU_CHUNK=$1 GPIRT_PERSIST_WGS=$2 GPIRT_POTRF_AUX=$3 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/scan.json 2> gpurun_out/scan.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/scan.json'));print('$cfg', round(d['value'],2), round(d['phases_ms']['potrf'],2), round(d['phases_ms']['nu'],2), d['substeps']['potrf']['rows_of_nu_computed_beside_it'], d['substeps']['potrf']['lookahead'])"
done
