"""Largest difference of f* (and of the other state arrays) after two sweeps at the benchmark size
between the three routes of the f* step: low-rank identity (default), triangular solves with
the interpolated cross-covariance (FSTAR_SOLVE), exact cross-covariance (EXACT_FSTAR)."""
import numpy as np

from gpirt_amd import _abi
from gpirt_amd.sampler import Sampler
from gpirt_amd.synth import generate

N, M, C = 10000, 4096, 5
d = generate(N, M, C, seed=1)
out = {}
for name, flags in (("lowrank", 0), ("solve", _abi.FSTAR_SOLVE), ("exact", _abi.EXACT_FSTAR)):
    cfg = _abi.default_config(n=N, m=M, H=1, C=C, seed=3, flags=flags)
    with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
        s.sweep(2)
        out[name] = s.state()
for a, b in (("lowrank", "exact"), ("solve", "exact"), ("lowrank", "solve")):
    print(a, "vs", b, {k: float(np.abs(out[a][k] - out[b][k]).max()) for k in ("fstar", "f", "theta", "beta")})
