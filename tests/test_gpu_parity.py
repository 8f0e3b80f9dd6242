"""GPU parity: every deterministic device kernel, called through the C ABI, against the CPU
oracle on the same seeded inputs (tolerances stated per test), then the full sweep."""
import os

import numpy as np
import pytest

from gpirt_amd import _abi
from conftest import load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from gpirt_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def test_library_sees_a_device():
    from gpirt_amd import _lib
    assert _lib.lib().gpirt_device_count() >= 1


def test_rng_matches_oracle_stream(K, O):
    # same Philox counters; Box-Muller differs only by libm vs device-libm rounding
    for tag, unit in ((_abi.TAG_F_Z, 0), (_abi.TAG_T_Z, 4095)):
        z_g = K.rng_normal(12345, tag, 7, unit, 4097)
        z_c = O.rng_normal(12345, tag, 7, unit, 4097)
        assert np.abs(z_g - z_c).max() < 1e-13
    u_g = K.rng_uniform(12345, _abi.TAG_F_U, 7, 3, 1001)
    assert np.array_equal(u_g, O.rng_uniform(12345, _abi.TAG_F_U, 7, 3, 1001))   # integer path: bit-exact


@pytest.mark.parametrize("n", [1, 100, 128, 300])
def test_k_build(K, O, n):
    x = np.random.default_rng(n).normal(size=n)
    a, b = K.k_build(x, 1.3, 0.7, 1e-6), O.k_build(x, 1.3, 0.7, 1e-6)
    assert np.abs(a - b).max() < 1e-15      # one exp per entry


@pytest.mark.parametrize("n", [5, 128, 300, 1000])
def test_potrf(K, O, n):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    S = A @ A.T + n * np.eye(n)              # well conditioned: compare factors entrywise
    Lg, Lc = K.potrf(S), O.potrf(S)
    assert np.abs(np.triu(Lg, 1)).max() == 0
    assert np.abs(Lg - Lc).max() < 1e-10 * np.abs(Lc).max()
    # the sampler's own matrix (RBF + 1e-6 I, condition ~1e9): backward error only
    x = rng.normal(size=n)
    Kx = O.k_build(x)
    Lk = K.potrf(Kx)
    assert np.abs(Lk @ Lk.T - Kx).max() < 1e-13 * n * np.abs(Kx).max()


def test_potrf_reports_not_pd(K):
    from gpirt_amd._lib import GpirtError
    S = np.eye(200)
    S[150, 150] = -1.0
    with pytest.raises(GpirtError) as e:
        K.potrf(S)
    assert e.value.code == _abi.GPIRT_ERR_NOT_PD


@pytest.mark.parametrize("n,k", [(100, 10), (300, 130), (640, 3)])
def test_trmm_lz(K, O, n, k):
    rng = np.random.default_rng(n + k)
    L = np.tril(rng.normal(size=(n, n)))     # asymmetric data: catches a transposed tile map
    Z = rng.normal(size=(n, k))
    a, b = K.trmm_lz(L, Z), O.trmm_lz(L, Z)
    assert np.abs(a - b).max() < 1e-12 * np.sqrt(n) * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("n,k", [(100, 7), (300, 140)])
def test_potrs(K, O, n, k):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    S = A @ A.T + n * np.eye(n)
    L = O.potrf(S)
    B = rng.normal(size=(n, k))
    a, b = K.potrs(L, B), O.potrs(L, B)
    assert np.abs(a - b).max() < 1e-11 * np.abs(b).max()
    assert np.abs(S @ a - B).max() < 1e-10 * np.abs(B).max()


@pytest.mark.parametrize("C", [2, 3, 5, 16])
def test_ordinal_loglik(K, O, C):
    rng = np.random.default_rng(C)
    n, u = 1000, 37
    b = np.zeros((C + 1, u))
    b[1:C] = np.sort(rng.normal(size=(C - 1, u)), axis=0)
    b[0], b[C] = -np.inf, np.inf
    g = 3.0 * rng.normal(size=(n, u))        # reaches both tails and the floor
    y = rng.integers(0, C + 1, size=(n, u)).astype(np.int8)
    (lu, lr), (ou, orow) = K.ordinal_loglik(g, y, b), O.ordinal_loglik(g, y, b)
    assert np.allclose(lu, ou, rtol=1e-12, atol=1e-10)
    assert np.allclose(lr, orow, rtol=1e-12, atol=1e-10)


def test_ordinal_loglik_all_missing(K):
    b = np.array([[-np.inf], [0.0], [np.inf]])
    lu, lr = K.ordinal_loglik(np.zeros((70, 1)), np.zeros((70, 1), dtype=np.int8), b)
    assert lu[0] == 0 and np.all(lr == 0)


@pytest.mark.parametrize("C,n", [(2, 100), (5, 1000), (3, 2100)])   # 2100 rows: workgroup-per-unit path
def test_ess_f_injected_streams(K, O, C, n):
    """Same f, nu, uniforms => identical accept/shrink decisions and the same new f."""
    rng = np.random.default_rng(C * n)
    u, R = 23, 40
    f = rng.normal(size=(n, u)); nu = rng.normal(size=(n, u)); mu = 0.5 * rng.normal(size=(n, u))
    y = rng.integers(0, C + 1, size=(n, u)).astype(np.int8)
    b = np.zeros((C + 1, u)); b[1:C] = np.sort(rng.normal(size=(C - 1, u)), axis=0)
    b[0], b[C] = -np.inf, np.inf
    un = rng.uniform(size=(R + 2, u))
    fg, rg, lg = K.ess_f(f, nu, mu, y, b, un, R)
    fc, rc, lc = O.ess_f(f, nu, mu, y, b, un, R)
    assert np.array_equal(rg, rc)
    assert rg.max() > 1                       # the shrink branch was exercised
    assert np.abs(fg - fc).max() < 1e-13
    assert np.allclose(lg, lc, rtol=1e-12)


def test_ess_f_round_cap(K, O):
    # an impossible slice height (u0 -> log ~ 0 and a likelihood that only gets worse is not
    # constructible in general), so cap at 1 round and check the kept state is untouched
    rng = np.random.default_rng(9)
    n, u, C, R = 64, 8, 2, 1
    f = rng.normal(size=(n, u)); nu = 5 * rng.normal(size=(n, u)); mu = np.zeros((n, u))
    y = (1 + (f > 0)).astype(np.int8)         # data agree with f: most first proposals fail
    b = np.tile(np.array([[-np.inf], [0.0], [np.inf]]), (1, u))
    un = np.vstack([np.full((1, u), 0.999999), rng.uniform(size=(R + 1, u))])
    fg, rg, _ = K.ess_f(f, nu, mu, y, b, un, R)
    fc, rc, _ = O.ess_f(f, nu, mu, y, b, un, R)
    assert np.array_equal(rg, rc) and (rg < 0).any()
    kept = rg < 0
    assert np.array_equal(fg[:, kept], f[:, kept])


def test_fstar_predict(K, O):
    rng = np.random.default_rng(11)
    n, u = 300, 12
    x = rng.normal(size=n)
    L = O.potrf(O.k_build(x, jitter=1e-4))
    f = L @ rng.normal(size=(n, u))
    a = K.fstar_predict(x, L, f)
    b = O.fstar_predict(x, L, f)
    # K + 1e-4 I has condition ~1e7: the two solve orders agree to ~cond * eps * |f|
    assert np.abs(a - b).max() < 1e-7 * max(1.0, np.abs(b).max())


def _pair(cfg, y, **kw):
    from gpirt_amd.sampler import Sampler
    from oracle import oracle
    return Sampler(cfg, y, **kw), oracle.Sampler(cfg, y, **kw)


def assert_states_agree(sg, sc, tag=None):
    """theta, beta and the thresholds are rotations of exactly shared draws once every accept /
    shrink decision coincides: they must agree to rounding (1e-9).  f and f* go through solves
    with K + 1e-6 I (condition ~1e9-1e10), which bounds their agreement at 1e-5."""
    for k in ("theta", "beta"):
        assert np.abs(sg[k] - sc[k]).max() < 1e-9, (tag, k)
    fin = np.isfinite(sc["threshold"])
    assert np.array_equal(fin, np.isfinite(sg["threshold"])), tag
    assert np.abs(sg["threshold"][fin] - sc["threshold"][fin]).max() < 1e-9, (tag, "threshold")
    for k in ("f", "fstar"):
        assert np.abs(sg[k] - sc[k]).max() < 1e-5, (tag, k)


@pytest.mark.parametrize("stem,C,const,flags", [("2PL_C_2", 2, 0, 0), ("Wiener_C_5", 5, 0, 0),
                                                 ("2PL_C_5", 5, 1, 0),
                                                 ("Wiener_C_5", 5, 0, _abi.EXACT_FSTAR),
                                                 ("2PL_C_5", 5, 1, _abi.EXACT_FSTAR),
                                                 ("Wiener_C_5", 5, 0, _abi.FSTAR_SOLVE),
                                                 ("2PL_C_5", 5, 1, _abi.FSTAR_SOLVE),
                                                 ("2PL_C_5", 5, 1, _abi.GENERATOR_PRIOR),
                                                 ("2PL_C_2", 2, 0, _abi.THETA_TABLE),
                                                 ("Wiener_C_5", 5, 0, _abi.THETA_TABLE),
                                                 ("2PL_C_5", 5, 1, _abi.THETA_TABLE),
                                                 ("Wiener_C_5", 5, 0, _abi.INDEX_TRUNCATE),
                                                 ("2PL_C_5", 5, 1, _abi.INDEX_TRUNCATE | _abi.THETA_TABLE)])
def test_sweep_matches_oracle_on_reference_fixtures(stem, C, const, flags):
    """Three whole sweeps from the same state with the shared generator: every accept /
    shrink decision must coincide, so the states agree to rounding (theta, beta, thresholds:
    1e-9; f and f*, which go through solves with K + 1e-6 I of condition ~1e9: 1e-5)."""
    d = load_fixture(stem)
    y = d["y"][:, :, :4] if const else d["y"]
    theta = d["theta"][:, :y.shape[2]]
    n, m, H = y.shape
    cfg = _abi.default_config(n=n, m=m, H=H, C=C, seed=5, theta_os=1.0, theta_ls=H / 2,
                              constant_irf=const, flags=flags)
    g, c = _pair(cfg, y, theta_init=theta)
    for sweep in range(3):
        g.sweep(1); c.sweep(1)
        assert_states_agree(g.state(), c.state(), sweep)
    dg, dc = g.diagnostics(), c.diagnostics()
    assert dg == dc
    g.close(); c.close()


@pytest.mark.parametrize("n,m,sweeps,flags", [(1300, 9, 2, 0), (2100, 5, 1, 0),
                                              (2100, 5, 1, _abi.EXACT_FSTAR),
                                              (2100, 5, 1, _abi.FSTAR_SOLVE),
                                              (2100, 5, 1, _abi.GENERATOR_PRIOR),
                                              (1300, 9, 2, _abi.GENERATOR_PRIOR),
                                              (1300, 9, 2, _abi.THETA_DIRECT),
                                              (1300, 9, 2, _abi.INDEX_TRUNCATE),    # bucketed table rows
                                              (1100, 2100, 1, 0)])   # two item chunks per table row
def test_sweep_matches_oracle_wide_units(n, m, sweeps, flags):
    """n >= 1024 rows per unit switches the item kernels to one workgroup per unit and the
    Cholesky to several diagonal blocks (ragged last super-block); n >= 2048 also splits the
    K range of the f* product into partial sums; from 1024 respondents on the respondent step
    sums entries of the per-sweep table of log cell probabilities unless THETA_DIRECT asks for
    cell evaluations (the fixtures above force the table with THETA_TABLE): same parity bar."""
    from gpirt_amd.synth import generate
    d = generate(n, m, 3, seed=4)
    cfg = _abi.default_config(n=n, m=m, H=1, C=3, seed=8, flags=flags)
    g, c = _pair(cfg, d["y"], theta_init=d["theta"])
    for sweep in range(sweeps):
        g.sweep(1); c.sweep(1)
        assert_states_agree(g.state(), c.state(), sweep)
    assert g.diagnostics() == c.diagnostics()
    g.close(); c.close()


@pytest.mark.parametrize("flags", [0, _abi.EXACT_FSTAR, _abi.FSTAR_SOLVE, _abi.GENERATOR_PRIOR])
def test_sweep_matches_oracle_many_items(flags):
    """More unit columns than grid points (ML >= G_pad = 1024): with the exact cross-covariance
    the f* phase then solves S^-1 K(x, xs) instead of S^-1 F; with FSTAR_SOLVE it solves for the
    Chebyshev-node columns only; by default it takes the low-rank identity and solves nothing
    with the n x n factor.  Same parity bar."""
    from gpirt_amd.synth import generate
    d = generate(70, 1030, 3, seed=6)
    cfg = _abi.default_config(n=70, m=1030, H=1, C=3, seed=2, flags=flags)
    g, c = _pair(cfg, d["y"], theta_init=d["theta"])
    for sweep in range(2):
        g.sweep(1); c.sweep(1)
        assert_states_agree(g.state(), c.state(), sweep)
    assert g.diagnostics() == c.diagnostics()
    g.close(); c.close()


def test_grid_mode_respondent_step_matches_oracle():
    """Static respondent step (utils/sample_score.m:1-20) on the device: table + one-hot GEMM +
    inverse-CDF draw against the oracle's triple loop, with missing cells.  theta lives on the
    grid, so agreement is exact up to the rounding of lo + step*k."""
    from gpirt_amd.synth import generate
    for n, m, C, seed in ((300, 21, 5, 3), (150, 7, 2, 9)):
        d = generate(n, m, C, seed=seed)
        y = d["y"].copy()
        rng = np.random.default_rng(seed)
        y[rng.random(y.shape) < 0.15] = 0
        cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=seed, theta_mode=_abi.THETA_GRID)
        g, c = _pair(cfg, y, theta_init=d["theta"])
        for sweep in range(3):
            g.sweep(1); c.sweep(1)
            sg, sc = g.state(), c.state()
            assert np.abs(sg["theta"] - sc["theta"]).max() < 1e-12, sweep
            for k in ("beta", "f", "fstar"):
                assert np.abs(sg[k] - sc[k]).max() < 1e-5, (sweep, k)
        assert g.diagnostics() == c.diagnostics()
        g.close(); c.close()


def test_grid_mode_rejects_dynamic_theta():
    from gpirt_amd.sampler import Sampler
    from gpirt_amd._lib import GpirtError
    cfg = _abi.default_config(n=20, m=3, H=2, C=2, theta_mode=_abi.THETA_GRID)
    with pytest.raises(GpirtError):
        Sampler(cfg, np.ones((20, 3, 2), dtype=np.int8))


def test_gpirtmcmc_drop_in_recovers_theta():
    """The reference's own acceptance metric (gpirt-sythetic.R:257-260,275) on its own
    fixture, through the gpirtMCMC mirror; and the GPU chain agrees with the CPU chain."""
    from gpirt_amd.mcmc import gpirtMCMC
    from oracle import oracle
    d = load_fixture("2PL_C_2")
    y, theta = d["y"], d["theta"]
    n, m, H = y.shape
    rng = np.random.default_rng(1)
    init = theta + 0.05 * rng.normal(size=theta.shape)
    data = np.where(y == 0, np.nan, y.astype(float))
    res = gpirtMCMC(data, 200, 200, THIN=4, CHAIN=1, beta_prior_means=np.zeros((3, m)),
                    beta_prior_sds=np.ones((3, m)), theta_os=1, theta_ls=H / 2, theta_init=init,
                    KERNEL="RBF", thresholds=None, SEED=1, constant_IRF=0)[0]
    assert res["theta"].shape == (50, n, H) and res["fstar"].shape == (50, 1001, m, H)
    assert res["beta"].shape == (50, 3, m, H) and res["threshold"].shape == (50, m, 3, H)
    est = res["theta"].mean(axis=0)
    cors = [abs(np.corrcoef(est[:, h], theta[:, h])[0, 1]) for h in range(H)]
    assert np.mean(cors) > 0.9
    cfg = _abi.default_config(n=n, m=m, H=H, C=2, sample_iterations=200, burn_iterations=200,
                              thin=4, seed=1, theta_os=1.0, theta_ls=H / 2)
    cpu = oracle.run(cfg, y, theta_init=init, beta_prior_means=np.zeros((3, m)),
                     beta_prior_sds=np.ones((3, m)))
    est_c = cpu["theta"].mean(axis=2)
    assert np.corrcoef(est.ravel(), est_c.ravel())[0, 1] > 0.99
    assert res["diagnostics"]["ess_capped"] == 0


@pytest.mark.parametrize("const,quad", [(0, False), (1, True)])
def test_device_summaries_match_the_kept_draws(const, quad):
    """The running sums kept on the device (theta mean / sd, category probabilities and most
    probable category on the grid: gpirt-sythetic.R:111-133,151-214) against the same
    quantities computed from the returned draws."""
    from gpirt_amd import post
    from gpirt_amd.mcmc import gpirtMCMC
    d = load_fixture("Wiener_C_5")
    H = 3
    y, theta = d["y_train"][:, :, :H], d["theta"][:, :H]
    ch = gpirtMCMC(y, 24, 8, THIN=4, theta_os=1.0, theta_ls=1.5, theta_init=theta, SEED=3,
                   constant_IRF=const, summary=True, summary_quadratic=quad,
                   summary_drop_first_theta=bool(const))[0]
    sm = ch["summary"]
    ref = post.summarize_draws(ch, quadratic=quad, drop_first_theta=bool(const))
    assert sm["draws"] == 6 == ref["draws"]
    assert np.abs(sm["theta_mean"] - ref["theta_mean"]).max() < 1e-12
    assert np.abs(sm["theta_sd"] - ref["theta_sd"]).max() < 1e-10
    P = 1 if const else H
    assert sm["prob_mean"].shape == (1001, y.shape[1], P, 5)
    assert np.abs(sm["prob_mean"] - ref["prob_mean"][:, :, :P]).max() < 1e-12
    assert np.abs(sm["argmax_mean"] - ref["argmax_mean"][:, :, :P]).max() < 1e-12
    ev = post.evaluate(sm, d["y"][:, :, :H], d["train_idx"][:, :, :H], theta)
    if not const:        # the fixture was generated with period-specific response functions
        assert ev["mean_abs_cor_theta"] > 0.9


def test_config_c_senate_116_recovers_the_nominate_ordering():
    """The Senate configuration end to end: the 116th Senate roll-call cube (103 x 703, binary,
    built by gpirt_amd.builders.senate_cube) through the gpirtMCMC mirror from a random start;
    the posterior mean of theta must order the senators like NOMINATE's first dimension
    (the comparison of gpirt-senate.R:86-127)."""
    from gpirt_amd import post
    from gpirt_amd.mcmc import gpirtMCMC
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "senate_116.npz"))
    data, nom = d["data"], d["nominate_dim1"]
    ch = gpirtMCMC(data, 80, 80, THIN=4, beta_prior_sds=np.ones((2, data.shape[1])), theta_os=1.0,
                   theta_ls=10.0, SEED=1, return_f=False, return_fstar=False, summary=True)[0]
    ok = np.isfinite(nom)
    theta = ch["summary"]["theta_mean"][:, 0]
    assert abs(np.corrcoef(theta[ok], nom[ok])[0, 1]) > 0.9
    ev = post.evaluate(ch["summary"], data)
    assert ev["train_acc"] > 0.85                      # roll calls are well predicted by one dimension


def test_checkpoint_restore_continues_the_chain_bit_for_bit():
    from gpirt_amd.sampler import Sampler
    from gpirt_amd._lib import GpirtError
    d = load_fixture("Wiener_C_5")
    y, theta = d["y"][:, :, :3], d["theta"][:, :3]
    n, m, H = y.shape
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=21, theta_ls=1.5)
    a = Sampler(cfg, y, theta_init=theta)
    a.sweep(3)
    blob = a.checkpoint()
    a.sweep(3)
    want, want_diag = a.state(), a.diagnostics()
    a.close()
    b = Sampler(cfg, y, theta_init=np.zeros_like(theta))      # different start: all of it is overwritten
    b.restore(blob)
    b.sweep(3)
    got = b.state()
    for k in ("theta", "f", "fstar", "beta"):
        assert np.array_equal(got[k], want[k]), k
    assert b.diagnostics() == want_diag
    other = Sampler(_abi.default_config(n=n, m=m, H=H, C=5, seed=22, theta_ls=1.5), y, theta_init=theta)
    with pytest.raises(GpirtError):
        other.restore(blob)                                   # another seed: refused
    b.close(); other.close()


@pytest.mark.parametrize("flags", [0, _abi.GENERATOR_PRIOR])
def test_inputs_outside_the_node_range_take_the_dense_routes(flags):
    """The low-rank f* step and the generator form of the prior factor interpolate the kernel on
    the grid's range [-5, 5]; a sweep that starts with a theta outside it must notice and go
    through the n x n factor instead.  Two respondents start at 5.6 and -6.2: the first sweep
    runs the fallback, later ones (theta back inside) the default routes; parity bar unchanged."""
    from gpirt_amd.synth import generate
    d = generate(300, 7, 3, seed=12)
    theta = d["theta"].copy()
    theta[3, 0], theta[77, 0] = 5.6, -6.2
    cfg = _abi.default_config(n=300, m=7, H=1, C=3, seed=4, flags=flags)
    g, c = _pair(cfg, d["y"], theta_init=theta)
    for sweep in range(3):
        g.sweep(1); c.sweep(1)
        assert_states_agree(g.state(), c.state(), sweep)
    assert g.diagnostics() == c.diagnostics()
    g.close(); c.close()


@pytest.mark.parametrize("flags", [0, _abi.GENERATOR_PRIOR])
def test_short_length_scale_falls_back_to_the_dense_routes(flags):
    """With irf_ls = 0.12 the kernel on [-5, 5] has far more than 128 significant eigenvalues:
    the node representation is refused at set-up and every route is the dense one."""
    from gpirt_amd.synth import generate
    d = generate(200, 5, 2, seed=13)
    cfg = _abi.default_config(n=200, m=5, H=1, C=2, seed=6, irf_ls=0.12, flags=flags)
    g, c = _pair(cfg, d["y"], theta_init=d["theta"])
    for sweep in range(2):
        g.sweep(1); c.sweep(1)
        assert_states_agree(g.state(), c.state(), sweep)
    g.close(); c.close()


@pytest.mark.parametrize("flags", [_abi.FSTAR_SOLVE, 0])
def test_256_nodes_with_fewer_inputs_than_nodes(flags):
    """irf_ls = 0.4 needs 256 Chebyshev nodes for the cross-covariance; with n = 100 inputs
    (padded to 128) the [units x nodes] image of the f* step is wider than the unit columns it
    is parked in are long.  FSTAR_SOLVE goes through that image on every sweep, the default
    route whenever a theta leaves the grid's range."""
    from gpirt_amd.synth import generate
    d = generate(100, 9, 3, seed=21)
    theta = d["theta"].copy()
    if not flags:
        theta[5, 0] = 5.4
    cfg = _abi.default_config(n=100, m=9, H=1, C=3, seed=7, irf_ls=0.4, flags=flags)
    g, c = _pair(cfg, d["y"], theta_init=theta)
    for sweep in range(3):
        g.sweep(1); c.sweep(1)
        assert_states_agree(g.state(), c.state(), sweep)
    assert g.diagnostics() == c.diagnostics()
    g.close(); c.close()


def test_clustered_theta_escalates_the_jitter_on_both_prior_routes():
    """Near-duplicate inputs with a jitter far below rounding level: the factorisation fails, the
    dense route retries with ten times the jitter and counts it (as the oracle does), and the
    generator form of the factor hands such a sweep to the dense route instead of giving up."""
    from gpirt_amd.synth import generate
    d = generate(300, 6, 3, seed=5)
    theta = np.repeat(d["theta"][::10], 10, axis=0) + 1e-9 * np.arange(300)[:, None]
    for flags in (0, _abi.GENERATOR_PRIOR):
        cfg = _abi.default_config(n=300, m=6, H=1, C=3, seed=3, jitter=1e-17, flags=flags)
        g, c = _pair(cfg, d["y"], theta_init=theta)
        g.sweep(1); c.sweep(1)
        dg, dc = g.diagnostics(), c.diagnostics()
        assert dg["jitter_retries"] >= 1 and dc["jitter_retries"] >= 1
        assert np.isfinite(g.state()["theta"]).all()
        g.close(); c.close()


@pytest.mark.parametrize("n,m,H,const", [(1, 1, 1, 0), (5, 3, 2, 0), (100, 10, 10, 0), (129, 2, 1, 0),
                                         (513, 3, 1, 0), (7, 2, 40, 1), (100, 4, 6, 1)])
def test_create_and_one_sweep_at_ragged_sizes(n, m, H, const):
    """Set-up buffers are sized from padded counts (128-row tiles, 512-row super-blocks, one
    group per period): sizes below one tile, one past a tile and one past a super-block, with
    one and several periods, must create, sweep once and agree with the oracle."""
    rng = np.random.default_rng(n + m + H)
    y = rng.integers(0, 4, size=(n, m, H)).astype(np.int8)          # C = 3, zeros = missing
    theta = rng.normal(size=(n, H))
    cfg = _abi.default_config(n=n, m=m, H=H, C=3, seed=9, theta_ls=max(H / 2, 1.0), constant_irf=const)
    g, c = _pair(cfg, y, theta_init=theta)
    g.sweep(1); c.sweep(1)
    assert_states_agree(g.state(), c.state(), (n, m, H))
    assert g.diagnostics() == c.diagnostics()
    g.close(); c.close()


def test_gpirtmcmc_passes_flags_through():
    """The mirror's `flags` reach gpirt_config: the generator form of the prior factor through
    gpirt_run gives the chain of the dense route (constant IRF: one group of 400 stacked inputs)."""
    from gpirt_amd.mcmc import gpirtMCMC
    d = load_fixture("2PL_C_5")
    y, theta = d["y"][:, :, :4], d["theta"][:, :4]
    kw = dict(THIN=2, theta_os=1.0, theta_ls=2.0, theta_init=theta, SEED=2, constant_IRF=1)
    a = gpirtMCMC(y, 8, 4, **kw)[0]
    b = gpirtMCMC(y, 8, 4, flags=_abi.GENERATOR_PRIOR, **kw)[0]
    for k in ("theta", "f", "fstar", "beta"):
        assert np.abs(np.asarray(a[k]) - np.asarray(b[k])).max() < 1e-5, k
    da, db = dict(a["diagnostics"]), dict(b["diagnostics"])
    da.pop("seconds"); db.pop("seconds")
    assert da == db


def test_three_chains_mix_by_rhat_and_effective_size():
    """CHAIN = 3 through the gpirtMCMC mirror (gpirt_esm.R:74-100: independent chains, then
    rstan::Rhat of theta across them; gpirt-senate.R:130-166: coda::effectiveSize): the chains
    differ draw by draw, agree in distribution (R-hat near 1 for nearly every respondent and
    period) and mix (effective sizes well above a handful)."""
    from gpirt_amd import diag
    from gpirt_amd.mcmc import gpirtMCMC
    d = load_fixture("2PL_C_5")
    H = 4
    y, theta = d["y"][:, :, :H], d["theta"][:, :H]
    n, m, _ = y.shape
    chains = gpirtMCMC(y, 1000, 200, THIN=4, CHAIN=3, theta_os=1.0, theta_ls=H / 2, theta_init=theta,
                       SEED=1, constant_IRF=0, return_f=False, return_fstar=False)
    assert len(chains) == 3 and all(c["theta"].shape == (250, n, H) for c in chains)
    assert np.abs(chains[0]["theta"] - chains[1]["theta"]).max() > 1e-3          # independent streams
    dg = diag.theta_diagnostics(chains)
    assert dg["rhat"].shape == (n, H) and dg["ess"].shape == (n, H)
    # slice updates of theta given f move slowly (f and theta are strongly coupled), so the bar
    # is "no chain sits somewhere else", not "converged to three digits"
    assert np.median(dg["rhat"]) < 1.1 and (dg["rhat"] < 1.3).mean() > 0.95
    assert np.median(dg["ess"]) > 20
    again = gpirtMCMC(y, 1000, 200, THIN=4, CHAIN=2, theta_os=1.0, theta_ls=H / 2, theta_init=theta,
                      SEED=1, constant_IRF=0, return_f=False, return_fstar=False)
    assert np.array_equal(again[1]["theta"], chains[1]["theta"])                 # chain c repeats bit for bit
