"""The kernels that the defaults no longer reach stay under the same parity bar.

The library picks its GEMM and panel-solve kernels once per process (environment switches read at
first use), so each variant runs in a child process of its own: the LDS-staged GEMM for every
product (`GPIRT_GEMM_LDS=1`, the default for transposed-A products only), the LDS-staged small-tile
GEMM (`GPIRT_SMALL_DIRECT=0`) and the panel solve as a product with the inverted diagonal block
(`GPIRT_NO_SUBSTITUTION=1`), and the factorisation's steps in their step-by-step form over all rows
(`GPIRT_POTRF_STEPWISE=1`: diagonal block, panel solve, update as three launches per 128 columns; the
default takes the diagonal super-block first).  The child calls the parity tests of tests/test_gpu_parity.py as
plain functions."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _child(rank, env):
    os.environ.update(env)
    for p in (os.path.dirname(HERE), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gpirt_amd import _abi, kernels as K
    from oracle import oracle as O
    import test_gpu_parity as tp
    for n in (5, 300, 1000):
        tp.test_potrf(K, O, n)
    tp.test_trmm_lz(K, O, 300, 130)
    tp.test_trmm_lz(K, O, 640, 3)
    tp.test_potrs(K, O, 300, 140)
    tp.test_sweep_matches_oracle_on_reference_fixtures("2PL_C_5", 5, 1, _abi.FSTAR_SOLVE)
    tp.test_sweep_matches_oracle_wide_units(1300, 9, 2, 0)
    tp.test_sweep_matches_oracle_wide_units(2100, 5, 1, _abi.EXACT_FSTAR)


@pytest.mark.parametrize("env", [{"GPIRT_GEMM_LDS": "1"},
                                 {"GPIRT_SMALL_DIRECT": "0", "GPIRT_NO_SUBSTITUTION": "1"},
                                 {"GPIRT_POTRF_STEPWISE": "1"}],
                         ids=["lds_gemm", "lds_small_gemm_and_inverse_panel_solve", "stepwise_factorisation"])
def test_variant_kernels_match_the_oracle(env):
    import torch.multiprocessing as mp
    mp.spawn(_child, args=(env,), nprocs=1, join=True)
