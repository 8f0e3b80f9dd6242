"""Statistical sanity of the bench configuration (n = 10 000, m = 4096, C = 5): run the chain
from a perturbed start and compare the posterior mean of theta with the generating values.
argv: burn keep [flags], flags OR-ed into gpirt_config.flags (128 = GPIRT_GENERATOR_PRIOR)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from gpirt_amd import _abi
from gpirt_amd.sampler import Sampler
from gpirt_amd.synth import generate

n, m, C = 10000, 4096, 5
burn, keep = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 40)
d = generate(n, m, C, seed=1)
rng = np.random.default_rng(7)
cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=3, flags=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
s = Sampler(cfg, d["y"], theta_init=d["theta"] + 0.5 * rng.normal(size=d["theta"].shape))
t0 = time.perf_counter()
s.sweep(burn)
for _ in range(keep):
    s.sweep(1)
    s.accumulate()
dt = time.perf_counter() - t0
sm = s.summary()
th = sm["theta_mean"][:, 0]
print("%d sweeps in %.2f s; cor(theta_mean, theta_true) = %.5f; mean posterior sd = %.4f; rmse = %.4f"
      % (burn + keep, dt, np.corrcoef(th, d["theta"][:, 0])[0, 1], sm["theta_sd"].mean(),
         np.sqrt(np.mean((th - th.mean() - d["theta"][:, 0] + d["theta"][:, 0].mean()) ** 2))))
print(s.diagnostics())
