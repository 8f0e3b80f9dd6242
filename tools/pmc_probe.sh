#!/bin/bash
# counter passes over tools/sclk_probe (register-fed GEMM alone): where do the MFMA pipe's idle cycles go?
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_probe; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf $out/$name
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- $R/tools/sclk_probe 2 0.05 > $out/$name.log 2>&1 || { tail -3 $out/$name.log; return 1; }
  python3 - $out/$name <<'P'
import csv,glob,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm8" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print({k:round(v/len(n[k])) for k,v in acc.items()})
P
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE && \
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum
