"""nu = L Z beside the factorisation (GPIRT_NU_OVERLAP = number of super-panels whose rows of nu
are computed on a second stream while the Cholesky runs; -1 here: the sampler's own choice): the
chain must be the same bit for bit as without it; prints ms per sweep, the phase times and the
rows computed beside the factorisation for each setting.

    python tools/nu_overlap.py [n] [m] [sweeps] [settings, comma separated]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from gpirt_amd import _abi  # noqa: E402
from gpirt_amd.sampler import Sampler  # noqa: E402
from gpirt_amd.synth import generate  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
settings = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,-1,12").split(",")]

d = generate(n, m, 5, seed=1)
rng = np.random.default_rng(1)
theta_init = d["theta"] + 0.05 * rng.normal(size=d["theta"].shape)
ref = None
for k in settings:
    os.environ.pop("GPIRT_NU_OVERLAP", None)
    if k >= 0:
        os.environ["GPIRT_NU_OVERLAP"] = str(k)
    cfg = _abi.default_config(n=n, m=m, H=1, C=5, seed=1, theta_os=1.0, theta_ls=1.0)
    s = Sampler(cfg, d["y"], theta_init=theta_init, beta_prior_means=np.zeros((3, m)),
                beta_prior_sds=np.ones((3, m)))
    s.sweep(6)
    p0 = s.phase_times()
    t0 = time.perf_counter()
    s.sweep(sweeps)
    dt = (time.perf_counter() - t0) / sweeps
    p1 = s.phase_times()
    st = s.state(f=True, fstar=False)
    rows = s.nu_rows_beside_factor()
    s.close()
    ph = {q: round((p1[q][0] - p0[q][0]) / sweeps, 2) for q in p1}
    same = ""
    if ref is None:
        ref = st
    else:
        same = "theta identical: %s, f identical: %s" % (np.array_equal(ref["theta"], st["theta"]),
                                                          np.array_equal(ref["f"], st["f"]))
    print("overlap %2d (%d rows): %.2f ms/sweep (%.1f sweeps/s) %s %s" % (k, rows, 1e3 * dt, 1 / dt, ph, same), flush=True)
