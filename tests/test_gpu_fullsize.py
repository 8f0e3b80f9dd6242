"""The benchmark's sizes (n = 10 000 respondents, m = 4096 items, C = 5), where the CPU oracle
would need minutes per sweep: the device path is checked through properties that do not
depend on the size -- backward error of the factorisation against matrix-vector products,
linearity of the triangular product, residual of the solve, a checksum of checksums for the
two log-likelihood reductions, additivity over a split of the rows, and the invariants one
sweep must leave behind (utils/elliptical_slice.m:47-102, main.m:57-73)."""
import numpy as np
import pytest

from gpirt_amd import _abi

pytestmark = pytest.mark.gpu

N, M, C = 10000, 4096, 5


@pytest.fixture(scope="module")
def K():
    from gpirt_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def factor(K):
    """theta ~ N(0,1), S = K(theta, theta) + 1e-6 I, L = chol(S): main.m:29,67, sample_fs.m:2."""
    x = np.random.default_rng(1).normal(size=N)
    S = K.k_build(x, 1.0, 1.0, 1e-6)
    return x, S, K.potrf(S)


def test_kernel_matrix_entries(factor):
    x, S, _ = factor
    rows = np.random.default_rng(2).integers(0, N, size=64)
    want = np.exp(-0.5 * (x[rows, None] - x[None, :]) ** 2)
    want[np.arange(64), rows] += 1e-6
    assert np.abs(S[rows] - want).max() < 1e-15          # one exp per entry
    assert np.array_equal(S[rows], S[:, rows].T)         # symmetric bit for bit


def test_cholesky_backward_error(factor):
    """|S v - L (L^T v)| <= c n eps |S| |v| for random v: the factor reproduces the matrix
    to the backward error of a Cholesky factorisation although its condition is ~1e10."""
    _, S, L = factor
    assert np.all(np.diag(L) > 0)
    for j in range(0, N, 1250):                          # strictly upper part: exact zeros
        assert not L[j:j + 1250, j + 1250:].any()
        assert not np.triu(L[j:j + 1250, j:j + 1250], 1).any()
    V = np.random.default_rng(3).normal(size=(N, 6))
    R = S @ V - L @ (L.T @ V)
    scale = np.linalg.norm(S, 2 if N <= 2000 else "fro") * np.linalg.norm(V)
    assert np.linalg.norm(R) < 1e-14 * scale


def test_triangular_product_linearity_and_value(K, factor):
    """nu = L Z (elliptical_slice.m:64) for 96 columns: against numpy, and linear in Z."""
    _, _, L = factor
    rng = np.random.default_rng(4)
    Z1, Z2 = rng.normal(size=(N, 96)), rng.normal(size=(N, 96))
    A1, A2, A3 = K.trmm_lz(L, Z1), K.trmm_lz(L, Z2), K.trmm_lz(L, 0.5 * Z1 - 3.0 * Z2)
    want = L @ Z1
    tol = 1e-12 * np.sqrt(N) * max(1.0, np.abs(want).max())
    assert np.abs(A1 - want).max() < tol
    assert np.abs(A3 - (0.5 * A1 - 3.0 * A2)).max() < tol


def test_solve_residual(K, factor):
    """X = S^-1 B through the factor (the two solves of main.m:57-58): residual at the level
    of the backward error, i.e. relative to |S| |X|, not to |B| (the condition is ~1e10)."""
    _, S, L = factor
    B = S @ np.random.default_rng(5).normal(size=(N, 8))     # a right-hand side with a tame solution
    X = K.potrs(L, B)
    R = S @ X - B
    assert np.linalg.norm(R) < 1e-13 * np.linalg.norm(S, "fro") * np.linalg.norm(X)


def test_loglik_checksum_of_checksums_and_additivity(K):
    """Per-item and per-respondent sums of the same cells (log_like_fn.m:3-4 in the ordinal
    form of gpirt-sythetic.R:159-167) must have the same total, and masking the two halves of
    the rows in turn must split every per-item sum in two."""
    rng = np.random.default_rng(6)
    g = 1.5 * rng.normal(size=(N, M))
    y = rng.integers(0, C + 1, size=(N, M)).astype(np.int8)        # 0 = missing
    b = np.zeros((C + 1, M))
    b[1:C] = np.sort(rng.normal(size=(C - 1, M)), axis=0)
    b[0], b[C] = -np.inf, np.inf
    lu, lr = K.ordinal_loglik(g, y, b)
    assert np.isfinite(lu).all() and np.isfinite(lr).all()
    assert abs(lu.sum() - lr.sum()) < 1e-12 * abs(lu.sum())
    top, bottom = y.copy(), y.copy()
    top[N // 2:] = 0
    bottom[:N // 2] = 0
    lu_t, lr_t = K.ordinal_loglik(g, top, b)
    lu_b, lr_b = K.ordinal_loglik(g, bottom, b)
    assert np.allclose(lu_t + lu_b, lu, rtol=1e-12, atol=0)
    assert np.array_equal(lr_t[:N // 2], lr[:N // 2]) and not lr_t[N // 2:].any()
    assert np.array_equal(lr_b[N // 2:], lr[N // 2:]) and not lr_b[:N // 2].any()


@pytest.mark.parametrize("flags", [0, _abi.EXACT_FSTAR, _abi.THETA_DIRECT, _abi.FSTAR_SOLVE,
                                   _abi.GENERATOR_PRIOR])
def test_sweep_invariants(flags):
    """Two sweeps of the benchmark configuration.  Whatever was drawn: thresholds stay ordered
    with infinite ends, no slice update ran into the round cap, f was re-anchored on the grid
    curve at each respondent's new theta (main.m:69-73), and the variants of two steps lead to
    the same chain: low-rank identity / triangular solves with interpolated / exact
    cross-covariance in the f* step (main.m:57-58), table rows served per grid index / every
    cell evaluated in every round in the respondent step, dense / generator form of the prior
    factor."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate
    d = generate(N, M, C, seed=1)
    cfg = _abi.default_config(n=N, m=M, H=1, C=C, seed=3, flags=flags)
    with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
        s.sweep(2)
        st, dg = s.state(), s.diagnostics()
    th, f, fs, thr = st["theta"][:, 0], st["f"][:, :, 0], st["fstar"][:, :, 0], st["threshold"][:, :, 0]
    assert dg["ess_capped"] == 0 and dg["jitter_retries"] == 0
    assert np.isfinite(th).all()
    assert np.isneginf(thr[:, 0]).all() and np.isposinf(thr[:, C]).all()
    assert (np.diff(thr[:, 1:C], axis=1) > 0).all()
    idx = np.clip(np.rint((th + 5.0) / 0.01).astype(np.int64), 0, 1000)   # int64() rounds, main.m:70
    near = np.abs(f - fs[idx]).max(axis=1)
    # the re-anchoring is a gather: exact wherever the index computed here is the device's
    # (a theta within rounding of a grid point may fall on either side)
    assert (near == 0).mean() > 0.999
    assert np.corrcoef(th, d["theta"][:, 0])[0, 1] > 0.9        # one early draw, started at the truth
    key = ("theta", "f", "fstar", "beta")
    if flags == 0:
        test_sweep_invariants.first = {k: st[k] for k in key}
    elif getattr(test_sweep_invariants, "first", None) is not None:
        for k in key:
            # same decisions, values equal to the accuracy of the solves with K + 1e-6 I
            assert np.abs(st[k] - test_sweep_invariants.first[k]).max() < 1e-5, k


def test_rows_of_nu_beside_the_factorisation_leave_the_chain_unchanged(monkeypatch):
    """nu = L Z (elliptical_slice.m:64, sample_fs.m:2): the rows whose part of L is final may be
    computed on a second stream while the factorisation goes on, and the factorisation may run
    the bulk of each trailing update on that stream beside the next block columns' steps
    (look-ahead).  Same tiles, same arithmetic per tile: six sweeps give the same chain bit for
    bit with the rows of 12 super-panels computed that way (forced), with look-ahead forced (with
    and without those rows), with the sampler's own choice (it times the schedules in its first
    sweeps), and with the plain schedule (GPIRT_NO_NU_OVERLAP, GPIRT_LOOKAHEAD=0)."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate
    d = generate(N, M, C, seed=1)

    def chain(flags, forced, ahead):
        for name, v in (("GPIRT_NU_OVERLAP", forced), ("GPIRT_LOOKAHEAD", ahead)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, str(v))
        cfg = _abi.default_config(n=N, m=M, H=1, C=C, seed=3, flags=flags)
        with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
            s.sweep(6)
            return s.state(fstar=False), s.nu_rows_beside_factor(), s.factor_lookahead(), s.diagnostics()

    base, rows0, la0, dg0 = chain(_abi.NO_NU_OVERLAP, None, 0)
    assert rows0 == 0 and not la0
    for flags, forced, ahead, rows_ok, la_ok in ((0, 12, 0, (12 * 512,), (False,)),
                                                 (0, 12, 1, (12 * 512,), (True,)),
                                                 (_abi.NO_NU_OVERLAP, None, 1, (0,), (True,)),
                                                 (0, None, None, (0, 14 * 512, 17 * 512), (False, True))):
        st, rows, la, dg = chain(flags, forced, ahead)
        assert rows in rows_ok and la in la_ok, (flags, forced, ahead, rows, la)
        assert dg == dg0
        for k in ("theta", "f", "beta"):
            assert np.array_equal(base[k], st[k]), (k, flags, forced, ahead)


def test_slabs_of_nu_beside_the_factorisation_leave_the_chain_unchanged(monkeypatch):
    """With a rank's share of the items nu = L Z has few column tiles and its K range runs in slabs side
    by side, added in slab order.  The rows of finished super-panels are computed that way too, beside the
    factorisation (same slabs, same order; slabs that hold nothing for those rows are left out): four
    sweeps give the same chain bit for bit with 12 or all 20 super-panels' rows forced there, with and
    without look-ahead, with the sampler's own choice, and with everything behind the factorisation."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate
    m = 512
    d = generate(N, m, C, seed=4)

    def chain(flags, forced, ahead):
        for name, v in (("GPIRT_NU_OVERLAP", forced), ("GPIRT_LOOKAHEAD", ahead)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, str(v))
        cfg = _abi.default_config(n=N, m=m, H=1, C=C, seed=6, flags=flags)
        with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
            s.sweep(4)
            return s.state(fstar=False), s.nu_rows_beside_factor(), s.factor_lookahead(), s.diagnostics()

    base, rows0, la0, dg0 = chain(_abi.NO_NU_OVERLAP, None, 0)
    assert rows0 == 0 and not la0
    for flags, forced, ahead, rows_ok, la_ok in ((0, 12, 0, (12 * 512,), (False,)),
                                                 (0, 12, 1, (12 * 512,), (True,)),
                                                 (0, 20, 1, ((N + 127) // 128 * 128,), (True,)),
                                                 (0, None, None, (0, 18 * 512), (False, True))):
        st, rows, la, dg = chain(flags, forced, ahead)
        assert rows in rows_ok and la in la_ok, (flags, forced, ahead, rows, la)
        assert dg == dg0
        for k in ("theta", "f", "beta"):
            assert np.array_equal(base[k], st[k]), (k, flags, forced, ahead)


def test_a_failed_factorisation_is_retried_the_same_way_under_every_schedule(monkeypatch):
    """K(theta, theta) + jitter I with a jitter far too small for near-duplicate inputs: the first
    attempts hit a non-positive pivot and the sampler retries with ten times the jitter
    (jitter_retries counts them).  With rows of nu beside the factorisation and with look-ahead, a
    failed attempt leaves launches on the second stream that still read and write the matrix: the
    retry must wait for them.  Same chain bit for bit as the plain schedule, same number of retries."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate
    n, m = 4300, 4096
    d = generate(n, m, C, seed=2)
    theta = np.round(d["theta"], 1)                 # ~60 distinct values: the kernel matrix has rank ~60

    def chain(flags, forced, ahead):
        for name, v in (("GPIRT_NU_OVERLAP", forced), ("GPIRT_LOOKAHEAD", ahead)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, str(v))
        cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=5, flags=flags)
        cfg.jitter = 1e-15
        with Sampler(cfg, d["y"], theta_init=theta) as s:
            s.sweep(2)
            return s.state(fstar=False), s.diagnostics()

    base, dg0 = chain(_abi.NO_NU_OVERLAP, None, 0)
    assert dg0["jitter_retries"] > 0
    for flags, forced, ahead in ((0, 5, 0), (0, 5, 1), (_abi.NO_NU_OVERLAP, None, 1)):
        st, dg = chain(flags, forced, ahead)
        assert dg == dg0, (flags, forced, ahead)
        for k in ("theta", "f", "beta"):
            assert np.array_equal(base[k], st[k]), (k, flags, forced, ahead)


def test_slice_steps_in_two_shares_leave_the_chain_unchanged(monkeypatch):
    """The slice updates of f and of (beta, thresholds) (elliptical_slice.m:47-102, one update per item) may run
    in two shares of the units, the second on a stream of its own beside the first share's next step, with the f*
    step (main.m:57-58, which reads f only) beside the second share's parameter update.  Every unit's update is the
    same arithmetic on the same random numbers wherever it runs: four sweeps give the same chain bit for bit with
    one launch per step (GPIRT_SLICE_SHARES=0), with the sampler's own split, with a ragged split and with all
    units but one in the second share; the same at a size where a unit is a single wave (no split there)."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate
    for n, m in ((4000, 2304), (600, 2048)):
        d = generate(n, m, C, seed=7)

        def chain(shares):
            if shares is None:
                monkeypatch.delenv("GPIRT_SLICE_SHARES", raising=False)
            else:
                monkeypatch.setenv("GPIRT_SLICE_SHARES", str(shares))
            cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=8)
            with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
                s.sweep(4)
                return s.state(fstar=True), s.diagnostics()

        base, dg0 = chain(0)
        for shares in (None, 777, m - 1, 10 * m):
            st, dg = chain(shares)
            assert dg == dg0, (n, shares)
            for k in ("theta", "f", "beta", "threshold", "fstar"):
                assert np.array_equal(base[k], st[k]), (k, n, shares)


def test_two_shares_beside_a_host_programs_own_streams(monkeypatch):
    """The same in a process that holds more HIP streams than the runtime has hardware queues (four by default), of
    every priority and busy between the sweeps -- a host program's own: the sampler's low-priority stream then
    shares a queue with others.  Same chain bit for bit as with one launch per step."""
    import ctypes as ct
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate
    from gpirt_amd import _lib
    _lib.lib()                                   # brings the runtime the library itself uses into the process
    hip = ct.CDLL("libamdhip64.so.7")
    n, m = 4000, 2304
    d = generate(n, m, C, seed=7)

    def chain(shares, extra):
        if shares is None:
            monkeypatch.delenv("GPIRT_SLICE_SHARES", raising=False)
        else:
            monkeypatch.setenv("GPIRT_SLICE_SHARES", str(shares))
        cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=8)
        with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
            streams, buf = [], ct.c_void_p()
            if extra:
                lo, hi = ct.c_int(), ct.c_int()
                assert hip.hipDeviceGetStreamPriorityRange(ct.byref(lo), ct.byref(hi)) == 0
                assert hip.hipMalloc(ct.byref(buf), ct.c_size_t(1 << 20)) == 0
                for pr in [0] * 6 + [lo.value] * 2 + [hi.value] * 2:
                    st = ct.c_void_p()
                    assert hip.hipStreamCreateWithPriority(ct.byref(st), ct.c_uint(1), ct.c_int(pr)) == 0
                    streams.append(st)
            for it in range(4):
                for st in streams:
                    assert hip.hipMemsetAsync(buf, ct.c_int(it), ct.c_size_t(1 << 20), st) == 0
                s.sweep(1)
            shares_used = s.slice_second_share()
            for st in streams:
                assert hip.hipStreamSynchronize(st) == 0
                assert hip.hipStreamDestroy(st) == 0
            if extra:
                assert hip.hipFree(buf) == 0
            return s.state(fstar=True), shares_used

    base, none = chain(0, False)
    st, used = chain(None, True)
    assert none == 0 and used == 3 * m // 8
    for k in ("theta", "f", "beta", "threshold", "fstar"):
        assert np.array_equal(base[k], st[k]), k
