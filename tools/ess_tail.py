"""How uneven are the units of one slice launch?  Per-unit proposal counts of an f update at the
bench size (n = 10 000, a share of the items), from the chain's own state after a few sweeps, and
what a greedy in-order hand-out of those units to the chip's workgroup slots costs against a
perfectly even one.  usage: python tools/ess_tail.py [items]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpirt_amd import _abi, kernels as K
from gpirt_amd.sampler import Sampler
from gpirt_amd.synth import generate

m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n, C = 10000, 5
d = generate(n, m, C, seed=3)
cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=1)
with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
    s.sweep(6)
    st = s.state(fstar=False)
theta, f = st["theta"][:, 0], st["f"][:, :, 0]
beta, thr = st["beta"][:, :, 0], st["threshold"][:, :, 0]
t0 = time.time()
L = K.potrf(K.k_build(theta, 1.0, 1.0, 1e-6))
nu = L @ np.random.default_rng(0).normal(size=(n, m))
print("factor + draws: %.1f s" % (time.time() - t0))
mu = beta[0][None, :] + beta[1][None, :] * theta[:, None] + beta[2][None, :] * theta[:, None] ** 2
R = 100
unif = np.random.default_rng(1).uniform(size=(R + 2, m))
_, rounds, _ = K.ess_f(f, nu, mu, d["y"][:, :, 0], thr.T.copy(), unif, R)
cost = 1.0 + np.abs(rounds)
print("proposals per unit: mean %.2f  sd %.2f  max %d  (capped: %d)" % (np.abs(rounds).mean(), np.abs(rounds).std(), np.abs(rounds).max(), (rounds < 0).sum()))
for full_m, slots in ((4096, 1024), (4096, 768)):
    reps = full_m // m
    c = np.tile(cost, reps)
    np.random.default_rng(2).shuffle(c)
    free = np.zeros(slots)
    for x in c:                      # in-order hand-out to the slot that frees first
        i = np.argmin(free)
        free[i] += x
    print("%d units on %d slots: makespan %.1f passes, even share %.1f (tail %.0f %%)" %
          (full_m, slots, free.max(), c.sum() / slots, 100 * (free.max() / (c.sum() / slots) - 1)))
