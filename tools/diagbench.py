"""Micro-benchmark: Cholesky of a 1280 x 1280 matrix through the C ABI (10 diagonal-block
kernels per call); run under `rocprofv3 --kernel-trace --stats` to time single kernels."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from gpirt_amd import kernels as K  # noqa: E402

rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
A = rng.normal(size=(n, n))
S = A @ A.T + n * np.eye(n)
for _ in range(3):
    K.potrf(S)
