"""A long chain at the benchmark's size with the slice steps in two shares and with one launch per step:
the final states and the counters must agree bit for bit.  usage: python tools/soak_shares.py [SWEEPS]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpirt_amd import _abi            # noqa: E402
from gpirt_amd.sampler import Sampler  # noqa: E402
from gpirt_amd.synth import generate   # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n, m, C = 10000, 4096, 5
d = generate(n, m, C, seed=1)


def chain(shares):
    if shares is None:
        os.environ.pop("GPIRT_SLICE_SHARES", None)
    else:
        os.environ["GPIRT_SLICE_SHARES"] = str(shares)
    cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=3)
    with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
        t0 = time.time()
        done = 0
        while done < sweeps:
            k = min(100, sweeps - done)
            s.sweep(k)
            done += k
            print("shares", shares, "sweep", done, "%.1f s" % (time.time() - t0), "second share", s.slice_second_share(), flush=True)
        return s.state(fstar=True), s.diagnostics(), sweeps / (time.time() - t0)


a, da, ra = chain(None)
b, db, rb = chain(0)
print("sweeps/s", round(ra, 2), round(rb, 2), da, db)
print({k: bool(np.array_equal(a[k], b[k])) for k in a}, "counters equal", da == db)
