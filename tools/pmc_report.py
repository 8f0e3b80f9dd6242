"""Per-kernel means of the counter passes written by tools/pmc_collect.sh.
usage: python tools/pmc_report.py PMC_DIR OUT.json substring[=label] ...
For every kernel-name substring: launches, mean duration (kernel trace of the fetch pass), the mean
per launch of every counter, HBM bytes = 2 x FETCH_SIZE KB x 1024 (gfx950 tallies 128-byte read
requests at 64 bytes) + WRITE_SIZE KB x 1024, and the rate that goes with them."""
import csv
import glob
import json
import os
import sys


def counters(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault((r["Kernel_Name"], r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
            out[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return out


def durations(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return out


def main():
    base, dest, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
    config = {}
    if subs and subs[0].startswith("config:"):          # e.g. config:n=10000,m=4096
        config = {k: int(v) for k, v in (kv.split("=") for kv in subs[0][7:].split(","))}
        subs = subs[1:]
    passes = {p: counters(os.path.join(base, p)) for p in ("fetch", "write", "sq", "sq2", "lds") if os.path.isdir(os.path.join(base, p))}
    dur = durations(os.path.join(base, "fetch"))
    report = {"how": "tools/pmc_collect.sh: one rocprofv3 --pmc run per pass (FETCH_SIZE | WRITE_SIZE | 8 SQ + GRBM | 8 SQ | LDS array + VALU + GRBM) of "
                     "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants` (12 sweeps with the set-up "
                     "sweeps); means per launch; durations are those of the FETCH_SIZE pass (counter collection "
                     "serialises kernels and lowers the clock: compare with the un-profiled kernel_stats)",
              "correction": "HBM bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (gfx950 counts a 128-byte read "
                            "request as 64 bytes; writes are exact)",
              "config": config, "kernels": {}}
    for spec in subs:
        sub, _, label = spec.partition("=")
        row = {"match": sub}
        for pname, table in passes.items():
            acc, n = {}, 0
            for (kname, _), vals in table.items():
                if sub in kname:
                    n += 1
                    for c, v in vals.items():
                        acc[c] = acc.get(c, 0.0) + v
            row.setdefault("launches", n)
            for c, v in acc.items():
                row[c] = v / max(n, 1)
        ds = [v for k, vs in dur.items() if sub in k for v in vs]
        if ds:
            row["mean_duration_us_profiled"] = sum(ds) / len(ds)
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes_per_launch"] = 2.0 * row["FETCH_SIZE"] * 1024 + row["WRITE_SIZE"] * 1024
            if ds:
                row["hbm_GBps_profiled"] = row["hbm_bytes_per_launch"] / (row["mean_duration_us_profiled"] * 1e-6) / 1e9
        if "SQ_INSTS_VALU" in row and "SQ_WAVES" in row and row["SQ_WAVES"]:
            row["valu_insts_per_wave"] = row["SQ_INSTS_VALU"] / row["SQ_WAVES"]
        if "SQ_ACTIVE_INST_VALU" in row and row.get("SQ_WAVE_CYCLES"):
            row["valu_active_share_of_wave_cycles"] = row["SQ_ACTIVE_INST_VALU"] / row["SQ_WAVE_CYCLES"]
        if row.get("SQ_LDS_IDX_ACTIVE") and row.get("GRBM_GUI_ACTIVE"):
            cyc = row["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
            row["lds_array_busy_share"] = row["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc          # per CU
            row["lds_bank_conflict_share_of_lds_cycles"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
            if row.get("SQ_ACTIVE_INST_VALU"):
                row["valu_busy_share"] = row["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc  # per SIMD, 4 clocks a unit
        report["kernels"][label or sub] = row
    json.dump(report, open(dest, "w"), indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
