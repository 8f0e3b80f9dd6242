#!/bin/bash
# super-tile shapes of the XCD-aware order of the whole nu = L*Z launch: time and HBM traffic per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/st; mkdir -p $out
cp $R/gpirt_amd/libgpirt_hip.so /tmp/tree.so
export GPIRT_NU_OVERLAP=0 GPIRT_LOOKAHEAD=0
cd /tmp && export TMPDIR=/tmp
for lib in $LIBS; do
  cp $R/tools/_ab/$lib.so $R/gpirt_amd/libgpirt_hip.so
  timeout -k 10 150 python3 $R/bench.py --no-cpu-baseline --no-variants --steps 10 --warmup 2 > $out/$lib.json 2> $out/$lib.err || { echo FAILED $lib; tail -3 $out/$lib.err; break; }
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/${lib}_$c
    timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${lib}_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $out/${lib}_$c.log 2>&1 || { echo FAILED pmc $lib $c; tail -3 $out/${lib}_$c.log; break 2; }
  done
  python3 - $out $lib <<'P'
import csv,glob,json,sys,os
out,lib=sys.argv[1:3]
d=json.load(open(os.path.join(out,lib+".json")))
v={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    tot=n=0
    for f in glob.glob(os.path.join(out,lib+"_"+c,"**","*counter_collection.csv"),recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_direct_kernel<false, 4>" in r["Kernel_Name"] and r["Counter_Name"]==c:
                tot+=float(r["Counter_Value"]); n+=1
    v[c]=tot/max(n,1)
hbm=(2*v["FETCH_SIZE"]+v["WRITE_SIZE"])*1024
print(lib.ljust(10),"nu phase %.3f ms" % d["phases_ms"]["nu"], "%.2f sweeps/s" % d["value"], "HBM %.2f GB per launch" % (hbm/1e9))
P
done
cp /tmp/tree.so $R/gpirt_amd/libgpirt_hip.so
