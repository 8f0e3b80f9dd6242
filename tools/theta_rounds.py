"""Per-sweep totals (us) of the respondent-step kernels in a rocprofv3 results database
(argv[1]), last sweep, and the durations of its evaluation kernels in launch order."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end - start from kernels where name like '%theta%' order by start").fetchall()
sweeps, cur, prev = [], [], None
for name, start, dur in rows:
    if prev is not None and start - prev > 5e6:
        sweeps.append(cur)
        cur = []
    cur.append((name.split("(")[0].replace("void gpirt::", "").replace("gpirt::", ""), dur / 1e3))
    prev = start
sweeps.append(cur)
last = sweeps[-1]
tot = {}
for name, d in last:
    tot[name] = tot.get(name, 0.0) + d
print({k: round(v) for k, v in tot.items()})
for key in ("eval", "bucket_kernel"):
    print(key, [round(d) for name, d in last if key in name])
