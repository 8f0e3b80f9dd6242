"""Loader of the HIP library.  There is no CPU fallback: if ``libgpirt_hip.so`` is missing
or no gfx950 device is visible, every compute call raises."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgpirt_hip.so")
_lib = None

_dp = C.POINTER(C.c_double)
_bp = C.POINTER(C.c_int8)
_ip = C.POINTER(C.c_int32)

# name -> (restype, argtypes): every symbol include/gpirt.h declares
SYMBOLS = {
    "gpirt_version": (C.c_char_p, []),
    "gpirt_last_error": (C.c_char_p, []),
    "gpirt_device_count": (C.c_int, []),
    "gpirt_set_device": (C.c_int, [C.c_int]),
    "gpirt_device_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "gpirt_config_defaults": (None, [C.POINTER(_abi.GpirtConfig)]),
    "gpirt_run": (C.c_int, [C.POINTER(_abi.GpirtConfig), _bp, _dp, _dp, _dp, _dp, _dp, _dp,
                            C.POINTER(_abi.GpirtOutputs)]),
    "gpirt_sampler_create": (C.c_int, [C.POINTER(_abi.GpirtConfig), _bp, _dp, _dp, _dp, _dp, _dp, _dp,
                                       C.POINTER(C.c_void_p)]),
    "gpirt_sampler_sweep": (C.c_int, [C.c_void_p, C.c_int32]),
    "gpirt_sampler_get_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp]),
    "gpirt_sampler_phase_times": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int64)]),
    "gpirt_sampler_update_times": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int64), _dp]),
    "gpirt_sampler_diagnostics": (C.c_int, [C.c_void_p, C.POINTER(_abi.GpirtOutputs)]),
    "gpirt_sampler_ll_checksum": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "gpirt_sampler_accumulate": (C.c_int, [C.c_void_p, C.c_int32]),
    "gpirt_sampler_summary": (C.c_int, [C.c_void_p, C.POINTER(_abi.GpirtSummary)]),
    "gpirt_run_summary": (C.c_int, [C.POINTER(_abi.GpirtConfig), _bp, _dp, _dp, _dp, _dp, _dp, _dp,
                                    C.POINTER(_abi.GpirtOutputs), C.POINTER(_abi.GpirtSummary), C.c_int32]),
    "gpirt_sampler_checkpoint_bytes": (C.c_int64, [C.c_void_p]),
    "gpirt_sampler_checkpoint": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "gpirt_sampler_restore": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "gpirt_sampler_stream": (C.c_void_p, [C.c_void_p]),
    "gpirt_sampler_destroy": (None, [C.c_void_p]),
    "gpirt_k_build": (C.c_int, [_dp, C.c_int32, C.c_double, C.c_double, C.c_double, _dp]),
    "gpirt_potrf": (C.c_int, [_dp, C.c_int32, _dp]),
    "gpirt_potrf_noalloc": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "gpirt_trmm_lz": (C.c_int, [_dp, _dp, C.c_int32, C.c_int32, _dp]),
    "gpirt_potrs": (C.c_int, [_dp, _dp, C.c_int32, C.c_int32, _dp]),
    "gpirt_ordinal_loglik": (C.c_int, [_dp, _bp, _dp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]),
    "gpirt_ess_f": (C.c_int, [_dp, _dp, _dp, _bp, _dp, _dp, C.c_int32, C.c_int32, C.c_int32,
                              C.c_int32, _ip, _dp]),
    "gpirt_fstar_predict": (C.c_int, [_dp, _dp, _dp, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                      C.c_int32, C.c_double, C.c_double, _dp]),
    "gpirt_rng_normal": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, _dp]),
    "gpirt_rng_uniform": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, _dp]),
}


class GpirtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("gpirt error %d: %s" % (code, message))
        self.code = code


def loaded():
    """True once ``libgpirt_hip.so`` is in the process."""
    return _lib is not None


def lib():
    """The loaded ``libgpirt_hip.so``; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C gpirt_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback"
                % SO_PATH)
        l = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise GpirtError(rc, lib().gpirt_last_error().decode("utf-8", "replace"))
