"""ctypes mirror of ``include/gpirt.h`` (struct layouts, enums, callback types)."""
import ctypes as C

GPIRT_OK = 0
GPIRT_ERR_BAD_ARG = -1
GPIRT_ERR_NOT_PD = -2
GPIRT_ERR_ESS_COLLAPSE = -3
GPIRT_ERR_HIP = -4
GPIRT_ERR_COMM = -5
GPIRT_ERR_NO_DEVICE = -6
GPIRT_ERR_INTERRUPT = -7

KERNEL_RBF = 0
KERNEL_MATERN52 = 1
THETA_ESS = 0
THETA_GRID = 1
RETURN_F = 1
RETURN_FSTAR = 2
ALLREDUCE_ON_STREAM = 4
EXACT_FSTAR = 8
THETA_TABLE = 16
THETA_DIRECT = 32
FSTAR_SOLVE = 64
GENERATOR_PRIOR = 128
INDEX_TRUNCATE = 512
NO_NU_OVERLAP = 256
FIXED_ORDER_SUM = 1024
SUM_BLOCK = 512

SUMMARY_QUADRATIC, SUMMARY_SKIP_THETA, SUMMARY_DROP_FIRST_THETA = 1, 2, 4

# stream tags of the counter-based generator (same values in both backends)
TAG_F_Z, TAG_F_U, TAG_P_Z, TAG_P_U, TAG_T_Z, TAG_T_U, TAG_INIT = 1, 2, 3, 4, 5, 6, 7

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_int32, C.c_void_p)


class GpirtConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("n", C.c_int32),
        ("m", C.c_int32),
        ("H", C.c_int32),
        ("C", C.c_int32),
        ("sample_iterations", C.c_int32),
        ("burn_iterations", C.c_int32),
        ("thin", C.c_int32),
        ("chain", C.c_int32),
        ("seed", C.c_uint64),
        ("theta_os", C.c_double),
        ("theta_ls", C.c_double),
        ("kernel", C.c_int32),
        ("constant_irf", C.c_int32),
        ("jitter", C.c_double),
        ("irf_ls", C.c_double),
        ("irf_sf", C.c_double),
        ("n_grid", C.c_int32),
        ("grid_lo", C.c_double),
        ("grid_step", C.c_double),
        ("theta_mode", C.c_int32),
        ("max_ess_rounds", C.c_int32),
        ("flags", C.c_int32),
        ("item_begin", C.c_int32),
        ("item_end", C.c_int32),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("allreduce", ALLREDUCE_FN),
        ("allreduce_user", C.c_void_p),
        ("progress", PROGRESS_FN),
        ("progress_user", C.c_void_p),
    ]


class GpirtSummary(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("draws", C.c_int32),
        ("theta_mean", C.POINTER(C.c_double)),
        ("theta_sd", C.POINTER(C.c_double)),
        ("prob_mean", C.POINTER(C.c_double)),
        ("argmax_mean", C.POINTER(C.c_double)),
    ]


class GpirtOutputs(C.Structure):
    _fields_ = [
        ("theta", C.POINTER(C.c_double)),
        ("f", C.POINTER(C.c_double)),
        ("fstar", C.POINTER(C.c_double)),
        ("beta", C.POINTER(C.c_double)),
        ("threshold", C.POINTER(C.c_double)),
        ("loglik", C.POINTER(C.c_double)),
        ("ess_f_rounds", C.c_int64),
        ("ess_param_rounds", C.c_int64),
        ("ess_theta_rounds", C.c_int64),
        ("ess_capped", C.c_int64),
        ("jitter_retries", C.c_int32),
        ("nu_rows_beside_factor", C.c_int32),
        ("factor_lookahead", C.c_int32),
        ("nu_tail_panels", C.c_int32),
        ("slice_second_share", C.c_int32),
        ("seconds_total", C.c_double),
    ]


def default_config(**kw):
    """The defaults ``gpirt_config_defaults`` writes (gpirt-sythetic.R:60-94,
    utils/model.m:6), as a Python-side constructor so that the oracle wrapper does
    not need the HIP library."""
    c = GpirtConfig()
    c.struct_size = C.sizeof(GpirtConfig)
    c.H = 1
    c.C = 2
    c.sample_iterations = 500
    c.burn_iterations = 500
    c.thin = 1
    c.seed = 1
    c.theta_os = 1.0
    c.theta_ls = 1.0
    c.kernel = KERNEL_RBF
    c.jitter = 1e-6
    c.irf_ls = 1.0
    c.irf_sf = 1.0
    c.n_grid = 1001
    c.grid_lo = -5.0
    c.grid_step = 0.01
    c.theta_mode = THETA_ESS
    c.max_ess_rounds = 200
    c.world = 1
    for k, v in kw.items():
        if not hasattr(c, k):
            raise TypeError("unknown gpirt_config field %r" % k)
        setattr(c, k, v)
    return c
