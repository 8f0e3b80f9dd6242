"""numpy front-end of the CPU oracle (``oracle/libgpirt_oracle.so``).

TEST INFRASTRUCTURE: imported only by ``tests/``, ``__graft_entry__.smoke`` and the
``cpu_baseline`` leg of ``bench.py``.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from gpirt_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgpirt_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_bp = C.POINTER(C.c_int8)
_ip = C.POINTER(C.c_int32)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_sampler_loglik.restype = C.c_double
        _lib.orc_ordinal_cell.restype = C.c_double
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a, copy=False):
    if a is None:
        return None
    a = np.asarray(a, dtype=np.float64)
    return np.array(a, order="F", copy=True) if copy else np.asfortranarray(a)


def _i8(a):
    return np.asfortranarray(np.asarray(a, dtype=np.int8))


def rng_normal(seed, tag, sweep, unit, count):
    out = np.empty(count)
    lib().orc_rng_normal(C.c_uint64(seed), C.c_uint32(tag), C.c_uint32(sweep), C.c_uint32(unit),
                         C.c_int64(count), _d(out))
    return out


def rng_uniform(seed, tag, sweep, unit, count):
    out = np.empty(count)
    lib().orc_rng_uniform(C.c_uint64(seed), C.c_uint32(tag), C.c_uint32(sweep), C.c_uint32(unit),
                          C.c_int64(count), _d(out))
    return out


def philox_raw(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox_raw(c, k, o)
    return [int(v) for v in o]


def k_build(x, ls=1.0, sf=1.0, jitter=1e-6):
    x = _f64(x)
    n = x.shape[0]
    K = np.empty((n, n), order="F")
    lib().orc_k_build(_d(x), n, C.c_double(ls), C.c_double(sf), C.c_double(jitter), _d(K))
    return K


def potrf(S):
    S = _f64(S)
    n = S.shape[0]
    L = np.empty((n, n), order="F")
    rc = lib().orc_potrf(_d(S), n, _d(L))
    if rc:
        raise np.linalg.LinAlgError("oracle potrf: not positive definite")
    return L


def trmm_lz(L, Z):
    L, Z = _f64(L), _f64(Z)
    n, k = Z.shape
    N = np.empty((n, k), order="F")
    lib().orc_trmm_lz(_d(L), _d(Z), n, k, _d(N))
    return N


def potrs(L, B):
    L, B = _f64(L), _f64(B)
    n, k = B.shape
    X = np.empty((n, k), order="F")
    lib().orc_potrs(_d(L), _d(B), n, k, _d(X))
    return X


def ordinal_loglik(g, y, b):
    """g[n,u], y[n,u] int8, b[C+1,u] -> (ll_unit[u], ll_row[n])."""
    g, y, b = _f64(g), _i8(y), _f64(b)
    n, u = g.shape
    Cc = b.shape[0] - 1
    lu, lr = np.empty(u), np.empty(n)
    lib().orc_ordinal_loglik(_d(g), y.ctypes.data_as(_bp), _d(b), n, u, Cc, _d(lu), _d(lr))
    return lu, lr


def ordinal_cell(y, g, b):
    b = _f64(b)
    return lib().orc_ordinal_cell(int(y), C.c_double(g), _d(b))


def ess_f(f, nu, mu, y, b, unif, max_rounds):
    """Returns (f_new[n,u], rounds[u], ll[u])."""
    f = _f64(f, copy=True)
    nu, y, b, unif = _f64(nu), _i8(y), _f64(b), _f64(unif)
    mu = _f64(mu)
    n, u = f.shape
    Cc = b.shape[0] - 1
    assert unif.shape == (max_rounds + 2, u)
    rounds = np.empty(u, dtype=np.int32)
    ll = np.empty(u)
    lib().orc_ess_f(_d(f), _d(nu), _d(mu), y.ctypes.data_as(_bp), _d(b), _d(unif), n, u, Cc,
                    max_rounds, rounds.ctypes.data_as(_ip), _d(ll))
    return f, rounds, ll


def fstar_predict(x, L, f, ls=1.0, sf=1.0, n_grid=1001, grid_lo=-5.0, grid_step=0.01):
    x, L, f = _f64(x), _f64(L), _f64(f)
    n, u = f.shape
    out = np.empty((n_grid, u), order="F")
    lib().orc_fstar_predict(_d(x), _d(L), _d(f), n, u, C.c_double(ls), C.c_double(sf), n_grid,
                            C.c_double(grid_lo), C.c_double(grid_step), _d(out))
    return out


def num_threads():
    return lib().orc_num_threads()


class Inputs:
    """Host arrays in the ABI's layouts, kept alive for the duration of a call."""

    def __init__(self, cfg, y, theta_init=None, beta_prior_means=None, beta_prior_sds=None,
                 theta_prior_means=None, theta_prior_sds=None, thresholds=None):
        self.y = _i8(y).reshape((cfg.n, cfg.m, cfg.H), order="F")
        self.theta_init = _f64(theta_init)
        self.bpm = _f64(beta_prior_means)
        self.bps = _f64(beta_prior_sds)
        self.tpm = _f64(theta_prior_means)
        self.tps = _f64(theta_prior_sds)
        self.thr = _f64(thresholds)
        if self.thr is not None:
            self.thr = np.where(np.isfinite(self.thr), self.thr, 0.0)

    def args(self):
        return (self.y.ctypes.data_as(_bp), _d(self.theta_init), _d(self.bpm), _d(self.bps),
                _d(self.tpm), _d(self.tps), _d(self.thr))


def shapes(cfg):
    j1 = cfg.item_end if cfg.item_end > cfg.item_begin else cfg.m
    j0 = cfg.item_begin if cfg.item_end > cfg.item_begin else 0
    mloc = j1 - j0
    P = 1 if cfg.constant_irf else cfg.H
    return mloc, P


class Sampler:
    """Step-wise oracle chain (``orc_sampler_*``)."""

    def __init__(self, cfg, y, **kw):
        self.cfg = cfg
        self.inputs = Inputs(cfg, y, **kw)
        self._h = C.c_void_p()
        rc = lib().orc_sampler_create(C.byref(cfg), *self.inputs.args(), C.byref(self._h))
        if rc:
            raise RuntimeError("orc_sampler_create failed: %d" % rc)

    def sweep(self, k=1):
        rc = lib().orc_sampler_sweep(self._h, int(k))
        if rc:
            raise RuntimeError("orc_sampler_sweep failed: %d" % rc)

    def state(self):
        c = self.cfg
        mloc, P = shapes(c)
        th = np.empty((c.n, c.H), order="F")
        f = np.empty((c.n, mloc, c.H), order="F")
        fs = np.empty((c.n_grid, mloc, P), order="F")
        be = np.empty((3, mloc, P), order="F")
        tr = np.empty((mloc, c.C + 1, P), order="F")
        lib().orc_sampler_get_state(self._h, _d(th), _d(f), _d(fs), _d(be), _d(tr))
        return dict(theta=th, f=f, fstar=fs, beta=be, threshold=tr)

    def loglik(self):
        return lib().orc_sampler_loglik(self._h)

    def phase_seconds(self):
        out = np.zeros(7)
        lib().orc_sampler_phase_seconds(self._h, _d(out))
        return dict(zip(("factor", "nu", "ess_f", "param", "fstar", "theta", "snap"), out.tolist()))

    def diagnostics(self):
        o = _abi.GpirtOutputs()
        lib().orc_sampler_diagnostics(self._h, C.byref(o))
        return dict(ess_f_rounds=o.ess_f_rounds, ess_param_rounds=o.ess_param_rounds,
                    ess_theta_rounds=o.ess_theta_rounds, ess_capped=o.ess_capped,
                    jitter_retries=o.jitter_retries)

    def close(self):
        if self._h:
            lib().orc_sampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run(cfg, y, **kw):
    """One chain through ``orc_run``; returns dict of draws shaped like the HIP wrapper's."""
    inp = Inputs(cfg, y, **kw)
    mloc, P = shapes(cfg)
    thin = max(cfg.thin, 1)
    K = cfg.sample_iterations // thin
    total = cfg.sample_iterations + cfg.burn_iterations
    o = _abi.GpirtOutputs()
    th = np.empty((cfg.n, cfg.H, K), order="F")
    be = np.empty((3, mloc, P, K), order="F")
    tr = np.empty((mloc, cfg.C + 1, P, K), order="F")
    ll = np.empty(total)
    f = np.empty((cfg.n, mloc, cfg.H, K), order="F") if cfg.flags & _abi.RETURN_F else None
    fs = np.empty((cfg.n_grid, mloc, P, K), order="F") if cfg.flags & _abi.RETURN_FSTAR else None
    o.theta, o.beta, o.threshold, o.loglik = _d(th), _d(be), _d(tr), _d(ll)
    if f is not None:
        o.f = _d(f)
    if fs is not None:
        o.fstar = _d(fs)
    rc = lib().orc_run(C.byref(cfg), *inp.args(), C.byref(o))
    if rc:
        raise RuntimeError("orc_run failed: %d" % rc)
    return dict(theta=th, f=f, fstar=fs, beta=be, threshold=tr, loglik=ll,
                ess_f_rounds=o.ess_f_rounds, ess_param_rounds=o.ess_param_rounds,
                ess_theta_rounds=o.ess_theta_rounds, ess_capped=o.ess_capped,
                jitter_retries=o.jitter_retries)
