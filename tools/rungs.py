"""Throughput over the size rungs (n, m) = (1000, 512) ... (10000, 4096), C = 5, H = 1: one `bench.py`
run per rung (20 warm-up + 100 timed sweeps, the CPU leg beside it) in a child process each.
and at a rank's share of the items (n = 10 000, m = 2048 / 1024 / 512).
usage: python tools/rungs.py OUT.json"""
import json
import os
import subprocess
import sys

RUNGS = ((1000, 512), (2500, 1024), (5000, 2048), (10000, 4096))
SHARES = (2048, 1024, 512)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    dest = sys.argv[1]
    rows = []
    for n, m in RUNGS:
        cmd = [sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20",
               "--n", str(n), "--m", str(m), "--no-variants"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            sys.stderr.write(out.stderr[-2000:])
            sys.exit(out.returncode)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        cpu = d.get("cpu_baseline") or {}
        rows.append({"n": n, "m": m, "sweeps_per_s": d["value"], "ms_per_sweep": d["ms_per_step"],
                     "phases_ms": {k: round(v, 3) for k, v in d["phases_ms"].items() if v > 0.005},
                     "dense_tflops": d.get("dense_tflops"),
                     "rows_of_nu_beside_the_factorisation": d["substeps"]["potrf"].get("rows_of_nu_computed_beside_it"),
                     "cpu_sweeps_per_s": cpu.get("value"), "cpu_cores": cpu.get("cores"),
                     "cpu_phases_s": cpu.get("phases_s"),
                     "gpu_over_cpu": d["value"] / cpu["value"] if cpu.get("value") else None})
        print(json.dumps(rows[-1]), flush=True)
    # a rank's share of the items at the full number of respondents (2, 4, 8 ranks), exchange latency not included
    share = []
    for m in SHARES:
        cmd = [sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20",
               "--n", "10000", "--m", str(m), "--no-variants", "--no-cpu-baseline"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            sys.stderr.write(out.stderr[-2000:])
            sys.exit(out.returncode)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        share.append({"n": 10000, "items_on_the_rank": m, "ranks_it_stands_for": 4096 // m, "sweeps_per_s": d["value"],
                      "ms_per_sweep": d["ms_per_step"],
                      "phases_ms": {k: round(v, 3) for k, v in d["phases_ms"].items() if v > 0.005},
                      "lookahead": d["substeps"]["potrf"].get("lookahead"),
                      "rows_of_nu_beside_the_factorisation": d["substeps"]["potrf"].get("rows_of_nu_computed_beside_it")})
        print(json.dumps(share[-1]), flush=True)
    json.dump({"how": "tools/rungs.py: python bench.py --steps 100 --warmup 20 --n N --m M --no-variants, one process per rung; "
                      "rank_share: the same at n = 10 000 with a rank's share of the 4096 items, no CPU leg",
               "rungs": rows, "rank_share": share}, open(dest, "w"), indent=1)


if __name__ == "__main__":
    main()
