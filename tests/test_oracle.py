"""CPU tests of the oracle itself: it is pinned against independent implementations
(scipy, the published Philox known-answer vectors) because the reference holds no
expected-output vectors for the sampler (parity unpinned)."""
import numpy as np
import pytest
import scipy.linalg as sl
import scipy.special as sp

from gpirt_amd import _abi
from oracle import oracle as O
from conftest import load_fixture


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert O.philox_raw([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox_raw([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox_raw([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rng_moments_and_stream_independence():
    z = O.rng_normal(7, _abi.TAG_F_Z, 3, 11, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    u = O.rng_uniform(7, _abi.TAG_F_U, 3, 11, 200000)
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.005
    z2 = O.rng_normal(7, _abi.TAG_F_Z, 3, 12, 1000)
    assert abs(np.corrcoef(z[:1000], z2)[0, 1]) < 0.15
    # a prefix of a stream does not depend on how many draws are requested
    assert np.array_equal(O.rng_normal(7, 1, 3, 11, 17), z[:17])


def test_k_build_matches_sekernel():
    rng = np.random.default_rng(0)
    x = rng.normal(size=257)
    K = O.k_build(x, ls=1.3, sf=0.7, jitter=1e-6)
    ref = 0.49 * np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 1.3 ** 2) + 1e-6 * np.eye(257)
    assert np.abs(K - ref).max() < 1e-15


def test_potrf_trmm_potrs_vs_scipy():
    rng = np.random.default_rng(1)
    x = rng.normal(size=300)
    K = O.k_build(x)
    L = O.potrf(K)
    assert np.abs(np.triu(L, 1)).max() == 0
    assert np.abs(L @ L.T - K).max() < 1e-13 * np.abs(K).max() * 300
    A = rng.normal(size=(300, 300))
    S = A @ A.T + 300 * np.eye(300)          # well conditioned: factors comparable entrywise
    assert np.abs(O.potrf(S) - sl.cholesky(S, lower=True)).max() < 1e-11
    Z = rng.normal(size=(300, 5))
    assert np.abs(O.trmm_lz(L, Z) - L @ Z).max() < 1e-12
    B = rng.normal(size=(300, 4))
    Ls = O.potrf(S)
    X = O.potrs(Ls, B)
    assert np.abs(X - sl.cho_solve((Ls, True), B)).max() < 1e-12
    with pytest.raises(np.linalg.LinAlgError):
        O.potrf(-np.eye(4))


def _ref_loglik(g, y, b):
    out = np.zeros_like(g)
    n, u = g.shape
    for k in range(u):
        for i in range(n):
            if y[i, k] > 0:
                zlo, zhi = b[y[i, k] - 1, k] - g[i, k], b[y[i, k], k] - g[i, k]
                p = sp.ndtr(-zlo) - sp.ndtr(-zhi) if zlo > 0 else sp.ndtr(zhi) - sp.ndtr(zlo)
                out[i, k] = np.log(max(p, 1e-300))
    return out


@pytest.mark.parametrize("C", [2, 3, 5])
def test_ordinal_loglik_vs_scipy(C):
    rng = np.random.default_rng(C)
    n, u = 211, 6
    b = np.zeros((C + 1, u))
    b[1:C] = np.sort(rng.normal(size=(C - 1, u)), axis=0)
    b[0], b[C] = -np.inf, np.inf
    g = 2.5 * rng.normal(size=(n, u))
    y = rng.integers(0, C + 1, size=(n, u)).astype(np.int8)   # 0 = missing
    lu, lr = O.ordinal_loglik(g, y, b)
    ref = _ref_loglik(g, y, b)
    assert np.allclose(lu, ref.sum(0), rtol=1e-13, atol=1e-12)
    assert np.allclose(lr, ref.sum(1), rtol=1e-13, atol=1e-12)
    # agrees with the naive pnorm difference of gpirt-sythetic.R:159-167 where that is
    # well conditioned
    mask = (np.abs(g) < 1.5) & (y > 0)
    for i, k in zip(*np.nonzero(mask)):
        naive = np.log(sp.ndtr(b[y[i, k], k] - g[i, k]) - sp.ndtr(b[y[i, k] - 1, k] - g[i, k]))
        # the naive form carries an absolute error of ~1e-16 in p, i.e. 1e-16/p in log p
        assert abs(naive - ref[i, k]) < 1e-12 + 1e-15 / np.exp(ref[i, k])


def test_loglik_binary_is_log_like_fn():
    # C = 2 with threshold 0: P(y=2) = Phi(f), the Bernoulli form of utils/log_like_fn.m:3-4
    rng = np.random.default_rng(5)
    f = rng.normal(size=(50, 1))
    y01 = rng.integers(0, 2, size=(50, 1))
    b = np.array([[-np.inf], [0.0], [np.inf]])
    lu, _ = O.ordinal_loglik(f, (y01 + 1).astype(np.int8), b)
    ps = sp.ndtr(f)
    assert np.isclose(lu[0], np.sum(y01 * np.log(ps) + (1 - y01) * np.log(1 - ps)), rtol=1e-12)


def test_loglik_missing_and_empty():
    b = np.array([[-np.inf], [0.0], [np.inf]])
    lu, lr = O.ordinal_loglik(np.zeros((7, 1)), np.zeros((7, 1), dtype=np.int8), b)
    assert lu[0] == 0 and np.all(lr == 0)
    assert np.isfinite(O.ordinal_cell(1, 60.0, [-np.inf, 0.0, np.inf]))   # floored, not -inf


def test_ess_f_python_restatement():
    """The C loop against a line-by-line Python restatement of elliptical_slice.m:66-101."""
    rng = np.random.default_rng(3)
    n, u, C, R = 60, 5, 3, 30
    f = rng.normal(size=(n, u)); nu = rng.normal(size=(n, u)); mu = 0.3 * rng.normal(size=(n, u))
    y = rng.integers(0, C + 1, size=(n, u)).astype(np.int8)
    b = np.zeros((C + 1, u)); b[1:C] = np.sort(rng.normal(size=(C - 1, u)), axis=0)
    b[0], b[C] = -np.inf, np.inf
    un = rng.uniform(size=(R + 2, u))
    f1, rounds, ll = O.ess_f(f, nu, mu, y, b, un, R)
    for k in range(u):
        def L(v):
            return _ref_loglik((mu[:, k] + v)[:, None], y[:, [k]], b[:, [k]]).sum()
        hh = np.log(un[0, k]) + L(f[:, k])
        phi = un[1, k] * 2 * np.pi
        pmin, pmax, r = phi - 2 * np.pi, phi, 0
        while True:
            prop = f[:, k] * np.cos(phi) + nu[:, k] * np.sin(phi)
            l = L(prop); r += 1
            if l > hh:
                break
            if phi > 0: pmax = phi
            else: pmin = phi
            phi = un[1 + r, k] * (pmax - pmin) + pmin
        assert rounds[k] == r
        assert np.allclose(f1[:, k], prop, atol=1e-13)
        assert np.isclose(ll[k], l, rtol=1e-12)


def test_fstar_predict_vs_scipy():
    rng = np.random.default_rng(4)
    n, u = 80, 3
    x = rng.normal(size=n)
    K = O.k_build(x, jitter=1e-4)              # conditioned well enough to compare entrywise
    L = O.potrf(K)
    f = L @ rng.normal(size=(n, u))
    fs = O.fstar_predict(x, L, f, n_grid=101, grid_lo=-5.0, grid_step=0.1)
    xs = -5.0 + 0.1 * np.arange(101)
    Ks = np.exp(-0.5 * (xs[:, None] - x[None, :]) ** 2)
    ref = Ks @ sl.cho_solve((L, True), f)
    assert np.abs(fs - ref).max() < 1e-7


@pytest.mark.parametrize("stem,C", [("2PL_C_2", 2), ("Wiener_C_5", 5)])
def test_chain_recovers_theta_on_reference_fixtures(stem, C):
    """Stochastic acceptance on the reference's own synthetic inputs, with the settings of
    gpirt-sythetic.R:60-94 shortened: the posterior mean of theta must correlate with the
    generating theta (the driver prints this as mean |cor|, :275)."""
    d = load_fixture(stem)
    y, theta = d["y"], d["theta"]
    n, m, H = y.shape
    rng = np.random.default_rng(1)
    cfg = _abi.default_config(n=n, m=m, H=H, C=C, sample_iterations=120, burn_iterations=120,
                              thin=4, seed=1, theta_os=1.0, theta_ls=H / 2)
    r = O.run(cfg, y, theta_init=theta + 0.05 * rng.normal(size=theta.shape))
    est = r["theta"].mean(axis=2)
    cors = [abs(np.corrcoef(est[:, h], theta[:, h])[0, 1]) for h in range(H)]
    assert np.mean(cors) > 0.9
    assert r["ess_capped"] == 0 and r["jitter_retries"] == 0
    thr = r["threshold"]
    assert np.all(np.isneginf(thr[:, 0])) and np.all(np.isposinf(thr[:, C]))
    assert np.all(np.diff(thr[:, 1:C], axis=1) > 0)
    assert r["loglik"][-1] > r["loglik"][0]


def test_chain_is_deterministic_and_shard_invariant():
    """Same seed => same chain; and the generator is keyed by the global item index, so the
    items of a 2-way shard evolve exactly as in the unsharded run when theta is shared."""
    d = load_fixture("2PL_C_2")
    y = d["y"][:, :, :2]
    n, m, H = y.shape
    kw = dict(n=n, m=m, H=H, C=2, seed=3, theta_ls=1.0)
    a = O.Sampler(_abi.default_config(**kw), y, theta_init=d["theta"][:, :2])
    b = O.Sampler(_abi.default_config(**kw), y, theta_init=d["theta"][:, :2])
    a.sweep(2); b.sweep(2)
    sa, sb = a.state(), b.state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k])
