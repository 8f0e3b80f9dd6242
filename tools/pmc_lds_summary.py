"""Per-kernel means of the LDS counter pass (tools/pmc_collect.sh ... PMC_ONLY_LDS=1).
usage: python tools/pmc_lds_summary.py PMC_DIR  -- LDS-array and VALU busy shares of the kernel's cycles"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    if not any(s in k for s in ("ess_", "theta_eval", "theta_table")):
        continue
    n = len(cnt[k])
    cyc = c["GRBM_GUI_ACTIVE"] / 8 / n                      # per XCD
    lds = c["SQ_LDS_IDX_ACTIVE"] / 256 / n                  # per CU
    conf = c["SQ_LDS_BANK_CONFLICT"] / 256 / n
    valu = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / n          # per SIMD
    print("%-40s launches %3d  cycles %.2fM  LDS busy %.0f%% (conflicts %.0f%% of it)  VALU busy %.0f%%  VALU/LDS instr %.1f" %
          (k, n, cyc / 1e6, 100 * lds / cyc, 100 * conf / max(lds, 1), 100 * valu / cyc, c["SQ_INSTS_VALU"] / max(c["SQ_INSTS_LDS"], 1)))
