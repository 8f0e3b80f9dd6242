"""The TAPS configuration on the device: the TAPS economic panel (251 panelists x 6
questions x 41 waves, C = 5, a third of the cells missing) with one response function per item
shared by all waves (`constant_IRF = 1`: every (panelist, wave) pair is an input of the item's
GP, 10 291 of them) and exactly the arguments of TAPS/gpirt_TAPS.R:17-36 --
`beta_prior_sds` rows 1 and 3 switched off, `theta_prior_sds = [1; 0]`, `theta_ls = 12` --
for both temporal kernels the hold-out script runs (TAPS/gpirt_TAPS_holdout.R:89,
TAPS/onerun:23).  A cut of the panel that the CPU oracle finishes in seconds is compared
sweep by sweep; the whole panel goes through size-independent properties."""
import os

import numpy as np
import pytest

from gpirt_amd import _abi
from test_gpu_parity import _pair, assert_states_agree

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def taps():
    return np.load(os.path.join(GOLDEN, "taps_cube.npz"))["data"]


def taps_arguments(n, m):
    """TAPS/gpirt_TAPS.R:25-29: slope only (sd 3), constant theta prior term only."""
    bps = np.full((3, m), 3.0)
    bps[0] = 0.0
    bps[2] = 0.0
    tps = np.ones((2, n))
    tps[1] = 0.0
    return bps, tps


@pytest.mark.parametrize("kernel", [_abi.KERNEL_RBF, _abi.KERNEL_MATERN52])
@pytest.mark.parametrize("flags", [0, _abi.GENERATOR_PRIOR])
def test_taps_cut_matches_oracle(kernel, flags):
    """First 60 panelists x 6 questions x 41 waves = 2460 stacked inputs per item: three sweeps
    against the oracle, every slice decision identical."""
    from gpirt_amd.builders import standardised_row_scores
    y = np.ascontiguousarray(taps()[:60])
    n, m, H = y.shape
    bps, tps = taps_arguments(n, m)
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=1, theta_os=1.0, theta_ls=12.0,
                              kernel=kernel, constant_irf=1, flags=flags)
    g, c = _pair(cfg, y, theta_init=standardised_row_scores(y), beta_prior_sds=bps,
                 theta_prior_sds=tps)
    for sweep in range(3):
        g.sweep(1); c.sweep(1)
        sg, sc = g.state(), c.state()
        assert_states_agree(sg, sc, sweep)
        # the switched-off terms stay at their prior mean exactly
        assert not sg["beta"][0].any() and not sg["beta"][2].any()
        assert np.abs(sg["beta"][1]).max() > 0
    assert g.diagnostics() == c.diagnostics()
    g.close(); c.close()


@pytest.mark.parametrize("kernel", [_abi.KERNEL_RBF, _abi.KERNEL_MATERN52])
@pytest.mark.parametrize("const", [0, 1])
def test_theta_trend_priors_match_oracle(kernel, const):
    """`theta_prior_means` / `theta_prior_sds` with both rows in use (constant and linear trend
    per respondent, gpirt_supremecourt.R:28-29, VAA/gpirt_VAA.R:131-132) on the reference's
    Wiener fixture, for both temporal kernels."""
    from conftest import load_fixture
    d = load_fixture("Wiener_C_5")
    H = 6
    y, theta = d["y_train"][:, :, :H], d["theta"][:, :H]
    n, m, _ = y.shape
    rng = np.random.default_rng(3)
    tpm = np.vstack([0.3 * rng.normal(size=n), 0.05 * rng.normal(size=n)])
    tps = np.vstack([np.full(n, 0.8), rng.uniform(0.0, 0.2, size=n)])
    tps[1, ::3] = 0.0
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_os=0.7, theta_ls=2.5, kernel=kernel,
                              constant_irf=const)
    g, c = _pair(cfg, y, theta_init=theta, theta_prior_means=tpm, theta_prior_sds=tps)
    for sweep in range(3):
        g.sweep(1); c.sweep(1)
        assert_states_agree(g.state(), c.state(), sweep)
    assert g.diagnostics() == c.diagnostics()
    g.close(); c.close()


def test_matern_and_rbf_trajectories_differ():
    """The KERNEL argument reaches the device: same seed, same data, different temporal kernel
    => a different chain (a silent fall-back to RBF would pass every other test)."""
    from gpirt_amd.sampler import Sampler
    y = np.ascontiguousarray(taps()[:40])
    n, m, H = y.shape
    out = []
    for kernel in (_abi.KERNEL_RBF, _abi.KERNEL_MATERN52):
        cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=1, theta_ls=12.0, kernel=kernel, constant_irf=1)
        with Sampler(cfg, y) as s:
            s.sweep(2)
            out.append(s.state(f=False, fstar=False)["theta"])
    assert np.abs(out[0] - out[1]).max() > 1e-3


@pytest.fixture(scope="module")
def full_panel_chains():
    """Four sweeps of the whole panel by the dense route and with the prior factor in generator
    form, TAPS/gpirt_TAPS.R's arguments."""
    from gpirt_amd.builders import standardised_row_scores
    from gpirt_amd.sampler import Sampler
    y = taps()
    n, m, H = y.shape
    bps, tps = taps_arguments(n, m)
    init = standardised_row_scores(y)
    res = {}
    for name, flags in (("dense", 0), ("generator", _abi.GENERATOR_PRIOR)):
        cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=1, theta_os=1.0, theta_ls=12.0,
                                  constant_irf=1, flags=flags)
        with Sampler(cfg, y, theta_init=init, beta_prior_sds=bps, theta_prior_sds=tps) as s:
            s.sweep(4)
            res[name] = (s.state(), s.diagnostics())
    return y, res


def test_full_panel_sweep_invariants(full_panel_chains):
    """251 x 6 x 41 with `constant_IRF = 1` (a 10 291-input GP per item, the size of the
    benchmark's factorisation): thresholds stay ordered with infinite ends, the switched-off
    coefficients stay zero, no slice update hits the round cap, f is the grid curve gathered at
    every (panelist, wave) input (main.m:69-73), missing cells included."""
    y, res = full_panel_chains
    n, m, H = y.shape
    for name, (st, dg) in res.items():
        th, f, fs, thr, beta = st["theta"], st["f"], st["fstar"], st["threshold"], st["beta"]
        assert dg["ess_capped"] == 0 and dg["jitter_retries"] == 0, name
        assert np.isfinite(th).all() and np.isfinite(f).all() and np.isfinite(fs).all()
        assert fs.shape == (1001, m, 1) and thr.shape == (m, 6, 1) and beta.shape == (3, m, 1)
        assert np.isneginf(thr[:, 0]).all() and np.isposinf(thr[:, 5]).all()
        assert (np.diff(thr[:, 1:5, 0], axis=1) > 0).all()
        assert not beta[0].any() and not beta[2].any()
        idx = np.clip(np.rint((th + 5.0) / 0.01).astype(np.int64), 0, 1000)     # [n, H]
        exact = 0
        for j in range(m):
            exact += (f[:, j, :] == fs[idx, j, 0]).sum()
        assert exact / (n * m * H) > 0.999, name


def test_full_panel_routes_agree(full_panel_chains):
    """Dense n x n factor against the generator form of the same factor: the same slice
    decisions and the same chain to what the condition of K + 1e-6 I leaves."""
    _, res = full_panel_chains
    (a, da), (b, db) = res["dense"], res["generator"]
    assert da == db
    for k in ("theta", "beta"):
        assert np.abs(a[k] - b[k]).max() < 1e-9, k
    for k in ("f", "fstar"):
        assert np.abs(a[k] - b[k]).max() < 1e-5, k


def test_full_panel_loglik_moves_up_from_the_crude_start():
    """From standardised row scores the chain's log-likelihood must rise: 30 sweeps through
    gpirt_run with the hold-out script's Matern kernel (TAPS/gpirt_TAPS_holdout.R:89)."""
    from gpirt_amd.builders import standardised_row_scores
    from gpirt_amd.mcmc import gpirtMCMC
    y = taps()
    n, m, H = y.shape
    bps, tps = taps_arguments(n, m)
    ch = gpirtMCMC(y, 10, 20, THIN=1, CHAIN=1, theta_init=standardised_row_scores(y),
                   beta_prior_sds=bps, theta_prior_sds=tps, theta_os=1, theta_ls=12,
                   thresholds=None, SEED=1, constant_IRF=1, KERNEL="Matern",
                   return_f=False, return_fstar=False, flags=_abi.GENERATOR_PRIOR)[0]
    ll = ch["loglik"]
    assert np.isfinite(ll).all()
    assert ll[-5:].mean() > ll[:3].mean()
    assert ch["theta"].shape == (10, n, H)
    assert ch["diagnostics"]["ess_capped"] == 0
