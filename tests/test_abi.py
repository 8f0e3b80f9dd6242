"""The C-ABI library loads on a machine without a GPU, exports every symbol that
include/gpirt.h declares, and refuses to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gpirt_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gpirt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(gpirt_[a-z_0-9]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_every_declared_symbol_is_exported_and_bound():
    lib = C.CDLL(_lib.SO_PATH)
    decl = declared_functions()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), "libgpirt_hip.so does not export %s" % name
        assert name in _lib.SYMBOLS, "gpirt_amd._lib does not bind %s" % name
    assert sorted(_lib.SYMBOLS) == decl


def test_collective_library_exports_what_its_header_declares():
    """include/gpirt_comm.h is served by libgpirt_rccl.so (RCCL, no torch)."""
    from gpirt_amd import comm
    text = open(os.path.join(ROOT, "include", "gpirt_comm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decl = sorted(set(re.findall(r"\b(gpirt_comm_[a-z_0-9]+)\s*\(", text)))
    lib = C.CDLL(comm.SO_PATH)
    assert len(decl) >= 8
    for name in decl:
        assert hasattr(lib, name), "libgpirt_rccl.so does not export %s" % name
    assert sorted(comm.SYMBOLS) == decl
    assert not any(n.startswith("gpirt_comm_") for n in _lib.SYMBOLS)     # the main library stays RCCL-free


def test_struct_layout_matches_header_defaults():
    lib = _lib.lib()
    c = _abi.GpirtConfig()
    lib.gpirt_config_defaults(C.byref(c))
    assert c.struct_size == C.sizeof(_abi.GpirtConfig)
    ref = _abi.default_config()
    for name, _ in _abi.GpirtConfig._fields_:
        if name in ("allreduce", "progress", "allreduce_user", "progress_user"):
            continue
        assert getattr(c, name) == getattr(ref, name), name
    assert lib.gpirt_version().startswith(b"gpirt-hip")


def test_flag_values_match_the_header():
    text = open(os.path.join(ROOT, "include", "gpirt.h")).read()
    flags = dict(re.findall(r"\bGPIRT_([A-Z_]+)\s*=\s*(\d+)\s*[,/\n]", text))
    for name in ("RETURN_F", "RETURN_FSTAR", "ALLREDUCE_ON_STREAM", "EXACT_FSTAR", "THETA_TABLE",
                 "THETA_DIRECT", "FSTAR_SOLVE", "GENERATOR_PRIOR", "NO_NU_OVERLAP", "INDEX_TRUNCATE", "FIXED_ORDER_SUM"):
        assert int(flags[name]) == getattr(_abi, name), name
    values = [getattr(_abi, n) for n in ("RETURN_F", "RETURN_FSTAR", "ALLREDUCE_ON_STREAM", "EXACT_FSTAR",
                                         "THETA_TABLE", "THETA_DIRECT", "FSTAR_SOLVE", "GENERATOR_PRIOR", "NO_NU_OVERLAP",
                                         "INDEX_TRUNCATE", "FIXED_ORDER_SUM")]
    assert len(set(values)) == len(values) and all(v & (v - 1) == 0 for v in values)   # distinct bits


def test_generator_prior_refuses_the_routes_that_need_the_dense_factor():
    cfg = _abi.default_config(n=10, m=3, H=1, C=2, flags=_abi.GENERATOR_PRIOR | _abi.FSTAR_SOLVE)
    h = C.c_void_p()
    y = np.ones((10, 3, 1), dtype=np.int8)
    rc = _lib.lib().gpirt_sampler_create(C.byref(cfg), y.ctypes.data_as(C.POINTER(C.c_int8)), None, None,
                                         None, None, None, None, C.byref(h))
    assert rc == _abi.GPIRT_ERR_BAD_ARG
    assert b"GENERATOR_PRIOR" in _lib.lib().gpirt_last_error()


@pytest.mark.skipif(_lib.lib().gpirt_device_count() > 0, reason="a GPU is visible")
def test_no_cpu_fallback_without_device():
    from gpirt_amd import kernels
    with pytest.raises(_lib.GpirtError) as e:
        kernels.k_build(np.zeros(4))
    assert e.value.code == _abi.GPIRT_ERR_NO_DEVICE
    from gpirt_amd.sampler import Sampler
    cfg = _abi.default_config(n=4, m=2, H=1, C=2)
    with pytest.raises(_lib.GpirtError):
        Sampler(cfg, np.ones((4, 2, 1), dtype=np.int8))


def test_bad_arguments_are_rejected_before_touching_the_device():
    from gpirt_amd.sampler import Sampler
    cfg = _abi.default_config(n=4, m=2, H=1, C=99)
    with pytest.raises(_lib.GpirtError) as e:
        Sampler(cfg, np.ones((4, 2, 1), dtype=np.int8))
    assert e.value.code == _abi.GPIRT_ERR_BAD_ARG
    cfg = _abi.default_config(n=4, m=2, H=1, C=2, world=2)   # world > 1 without a hook
    with pytest.raises(_lib.GpirtError) as e:
        Sampler(cfg, np.ones((4, 2, 1), dtype=np.int8))
    assert e.value.code == _abi.GPIRT_ERR_BAD_ARG
