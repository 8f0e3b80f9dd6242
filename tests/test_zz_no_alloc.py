"""The Cholesky entry point that allocates nothing (`gpirt_potrf_noalloc`, for callers that
cannot call hipMalloc) against `gpirt_potrf`, which allocates its work buffers per call: same
kernels on the same padded matrix, so the factors must be equal bit for bit.  Two calls in a
row share the library's static device array; the second is smaller, so whatever the first left
behind in the array must not leak into it.  Device buffers for the caller's side come from the
HIP runtime directly (ctypes), the same runtime the library is linked against."""
import ctypes as C

import numpy as np
import pytest

from gpirt_amd import _abi

pytestmark = pytest.mark.gpu


class Hip:
    """hipMalloc / hipMemcpy / hipFree of the runtime libgpirt_hip.so itself uses."""

    def __init__(self):
        from gpirt_amd import _lib
        _lib.lib()                                   # brings libamdhip64 into the process
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamSynchronize.argtypes = [C.c_void_p]

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), nbytes) == 0
        return p

    def upload(self, arr):
        arr = np.asfortranarray(arr, dtype=np.float64)
        p = self.alloc(arr.nbytes)
        assert self.rt.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1) == 0   # host -> device
        return p

    def download(self, p, shape):
        out = np.empty(shape, order="F")
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2) == 0   # device -> host
        return out

    def free(self, p):
        self.rt.hipFree(p)


def spd(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=n)
    return np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 1e-6 * np.eye(n)   # the sampler's own matrix


def noalloc_factor(hip, S, ld_extra=0):
    from gpirt_amd import _lib
    n = S.shape[0]
    lds = n + ld_extra
    buf = np.full((lds, n), np.nan, order="F")
    buf[:n] = S
    d_S = hip.upload(buf)
    d_L = hip.alloc(8 * lds * n)
    hip.rt.hipMemset(d_L, 0xFF, 8 * lds * n)
    rc = _lib.lib().gpirt_potrf_noalloc(d_S, lds, n, d_L, lds, None)
    L = hip.download(d_L, (lds, n))[:n] if rc == 0 else None
    hip.free(d_S); hip.free(d_L)
    return rc, L


def test_no_alloc_cholesky_matches_the_allocating_entry_point_twice_in_a_row():
    from gpirt_amd import kernels
    hip = Hip()
    for n, seed, extra in ((1000, 1, 0), (300, 2, 5)):      # the second call is smaller, ld > n
        S = spd(n, seed)
        want = kernels.potrf(S)
        rc, got = noalloc_factor(hip, S, extra)
        assert rc == 0
        assert np.array_equal(got, want), n
        assert not np.triu(got, 1).any()


def test_no_alloc_cholesky_in_place_and_status():
    from gpirt_amd import _lib
    hip = Hip()
    n = 200
    S = spd(n, 3)
    from gpirt_amd import kernels
    want = kernels.potrf(S)
    d = hip.upload(S)
    assert _lib.lib().gpirt_potrf_noalloc(d, n, n, d, n, None) == 0       # d_L may be d_S
    assert np.array_equal(hip.download(d, (n, n)), want)
    hip.free(d)
    bad = np.eye(n)
    bad[150, 150] = -1.0
    rc, _ = noalloc_factor(hip, bad)
    assert rc == _abi.GPIRT_ERR_NOT_PD
    rc, got = noalloc_factor(hip, S)                                      # the status word was cleared
    assert rc == 0 and np.array_equal(got, want)


def test_no_alloc_cholesky_refuses_sizes_beyond_its_static_array():
    from gpirt_amd import _lib
    hip = Hip()
    d = hip.alloc(8 * 16)
    rc = _lib.lib().gpirt_potrf_noalloc(d, 1025, 1025, d, 1025, None)
    assert rc == _abi.GPIRT_ERR_BAD_ARG
    assert b"1024" in _lib.lib().gpirt_last_error()
    hip.free(d)
