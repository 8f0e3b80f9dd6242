"""Start and end of the slice kernels of one sweep from a rocprofv3 kernel trace (us from the first one's start)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ess_f_kernel" in r["Kernel_Name"] or "ess_param_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for s0 in range(len(rows) - 3 * per, len(rows), per):
    t0 = int(rows[s0]["Start_Timestamp"])
    print(" | ".join("%s[%s] %.0f-%.0f" % (r["Kernel_Name"].split("::")[1][:9], r.get("Grid_Size_X", r.get("Grid_Size", "?")),
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3) for r in rows[s0:s0 + per]))
