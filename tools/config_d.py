"""The TAPS configuration on one GPU: the panel (251 panelists x 6 items x 41 waves) with one
response function per item shared by all waves (constant_IRF = 1: 10 291 stacked inputs), the
settings of TAPS/gpirt_TAPS.R:17-36.  Prints time per sweep and the phase split."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from gpirt_amd import _abi
from gpirt_amd.builders import standardised_row_scores
from gpirt_amd.sampler import Sampler

d = np.load(os.path.join(ROOT, "tests", "golden", "taps_cube.npz"))
y = d["data"]
n, m, H = y.shape
cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=1, theta_os=1.0, theta_ls=12.0, constant_irf=1,
                          flags=int(sys.argv[2]) if len(sys.argv) > 2 else 0)   # argv: sweeps [flags]
bps = np.full((3, m), 3.0); bps[0] = 0.0; bps[2] = 0.0
tps = np.ones((2, n)); tps[1] = 0.0
s = Sampler(cfg, y, theta_init=standardised_row_scores(y), beta_prior_sds=bps, theta_prior_sds=tps)
s.sweep(13)     # the sampler settles its schedule in its first sweeps
p0 = s.phase_times()
t0 = time.perf_counter()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s.sweep(K)
dt = (time.perf_counter() - t0) / K
p1 = s.phase_times()
print("config D: %.2f ms/sweep, %.1f sweeps/s" % (1e3 * dt, 1 / dt))
print({k: round((p1[k][0] - p0[k][0]) / K, 2) for k in p1})
print(s.diagnostics())
