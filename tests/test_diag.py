"""Chain diagnostics (gpirt_amd/diag.py) on closed-form cases: R and its coda / rstan packages
are not available to compare numbers with."""
import numpy as np

from gpirt_amd import diag


def _ar1(n, phi, rng):
    x = np.zeros(n)
    e = rng.normal(size=n)
    for t in range(1, n):
        x[t] = phi * x[t - 1] + e[t]
    return x


def test_effective_size_of_white_noise_and_ar1():
    rng = np.random.default_rng(0)
    n = 4000
    ess = np.mean([diag.effective_size(rng.normal(size=n)) for _ in range(10)])
    assert 0.85 * n < ess < 1.15 * n
    for phi in (0.5, 0.9):
        want = n * (1 - phi) / (1 + phi)                   # spectral density at zero of an AR(1)
        got = np.mean([diag.effective_size(_ar1(n, phi, rng)) for _ in range(10)])
        assert 0.8 * want < got < 1.25 * want, (phi, got, want)
    assert diag.effective_size(np.ones(50)) == 0.0
    assert diag.effective_size(rng.normal(size=(300, 3))).shape == (3,)


def test_yule_walker_recovers_an_ar2():
    rng = np.random.default_rng(1)
    n = 20000
    x = np.zeros(n)
    e = rng.normal(size=n)
    for t in range(2, n):
        x[t] = 0.6 * x[t - 1] - 0.3 * x[t - 2] + e[t]
    ar, var_pred = diag._ar_yule_walker(x)
    assert ar.size >= 2 and abs(ar[0] - 0.6) < 0.03 and abs(ar[1] + 0.3) < 0.03
    assert abs(var_pred - 1.0) < 0.05


def test_rhat_is_one_for_mixed_chains_and_large_for_separated_ones():
    rng = np.random.default_rng(2)
    same = rng.normal(size=(500, 3))
    assert 0.99 < diag.rhat(same) < 1.02
    shifted = same + np.array([0.0, 0.0, 3.0])
    assert diag.rhat(shifted) > 1.5
    scaled = same * np.array([1.0, 1.0, 6.0])               # same location, different spread: the
    assert diag.rhat(scaled) > 1.1                          # folded (tail) part catches it
    out = diag.theta_diagnostics([dict(theta=rng.normal(size=(80, 4, 2))) for _ in range(2)])
    assert out["ess"].shape == (4, 2) and out["rhat"].shape == (4, 2)
