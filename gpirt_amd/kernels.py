"""numpy front-end of the deterministic device kernels (the parity entry points of
``include/gpirt.h``).  Arrays are column-major; every call runs on the current HIP device."""
import ctypes as C

import numpy as np

from ._lib import lib, check, _dp, _bp, _ip


def _f(a, copy=False):
    a = np.asarray(a, dtype=np.float64)
    return np.array(a, order="F", copy=True) if copy else np.asfortranarray(a)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def k_build(x, ls=1.0, sf=1.0, jitter=1e-6):
    x = _f(x)
    n = x.shape[0]
    K = np.empty((n, n), order="F")
    check(lib().gpirt_k_build(_d(x), n, ls, sf, jitter, _d(K)))
    return K


def potrf(S):
    S = _f(S)
    n = S.shape[0]
    L = np.empty((n, n), order="F")
    check(lib().gpirt_potrf(_d(S), n, _d(L)))
    return L


def trmm_lz(L, Z):
    L, Z = _f(L), _f(Z)
    n, k = Z.shape
    N = np.empty((n, k), order="F")
    check(lib().gpirt_trmm_lz(_d(L), _d(Z), n, k, _d(N)))
    return N


def potrs(L, B):
    L, B = _f(L), _f(B)
    n, k = B.shape
    X = np.empty((n, k), order="F")
    check(lib().gpirt_potrs(_d(L), _d(B), n, k, _d(X)))
    return X


def ordinal_loglik(g, y, b):
    g, b = _f(g), _f(b)
    y = np.asfortranarray(np.asarray(y, dtype=np.int8))
    n, u = g.shape
    lu, lr = np.empty(u), np.empty(n)
    check(lib().gpirt_ordinal_loglik(_d(g), y.ctypes.data_as(_bp), _d(b), n, u, b.shape[0] - 1,
                                     _d(lu), _d(lr)))
    return lu, lr


def ess_f(f, nu, mu, y, b, unif, max_rounds):
    f = _f(f, copy=True)
    nu, mu, b, unif = _f(nu), _f(mu), _f(b), _f(unif)
    y = np.asfortranarray(np.asarray(y, dtype=np.int8))
    n, u = f.shape
    assert unif.shape == (max_rounds + 2, u)
    rounds = np.empty(u, dtype=np.int32)
    ll = np.empty(u)
    check(lib().gpirt_ess_f(_d(f), _d(nu), _d(mu), y.ctypes.data_as(_bp), _d(b), _d(unif), n, u,
                            b.shape[0] - 1, max_rounds, rounds.ctypes.data_as(_ip), _d(ll)))
    return f, rounds, ll


def fstar_predict(x, L, f, ls=1.0, sf=1.0, n_grid=1001, grid_lo=-5.0, grid_step=0.01):
    x, L, f = _f(x), _f(L), _f(f)
    n, u = f.shape
    out = np.empty((n_grid, u), order="F")
    check(lib().gpirt_fstar_predict(_d(x), _d(L), _d(f), n, u, ls, sf, n_grid, grid_lo, grid_step,
                                    _d(out)))
    return out


def rng_normal(seed, tag, sweep, unit, count):
    out = np.empty(count)
    check(lib().gpirt_rng_normal(seed, tag, sweep, unit, count, _d(out)))
    return out


def rng_uniform(seed, tag, sweep, unit, count):
    out = np.empty(count)
    check(lib().gpirt_rng_uniform(seed, tag, sweep, unit, count, _d(out)))
    return out
