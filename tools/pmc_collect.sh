#!/bin/bash
# Counter passes of one short bench run, one rocprofv3 run per pass (FETCH_SIZE and WRITE_SIZE do
# not fit one pass; the SQ pass holds 8 counters).  usage: tools/pmc_collect.sh OUTDIR [ENV=VAL ...]
# The program itself follows "--" (no env / shell wrapper there).
out=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
run() {  # name, counters...
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$out/$name -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $R/$out/$name.log 2>&1 || { tail -5 $R/$out/$name.log; exit 1; }
}
if [ -n "$PMC_ONLY_LDS" ]; then
  run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  exit $?
fi
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE && \
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA && \
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
