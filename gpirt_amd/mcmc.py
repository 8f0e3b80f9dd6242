"""``gpirtMCMC`` -- the drop-in entry point, mirroring the R call the reference drivers make
(gpirt-sythetic.R:88-94, gpirt-senate.R:77-81, TAPS/gpirt_TAPS.R:30-36,
VAA/gpirt_VAA.R:137-145).  Argument names, order and meaning follow those call sites."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib, check, _dp
from .sampler import HostInputs, local_shape


def responses_from_r(data):
    """R response array (numeric with NA=NaN, or integer with 0 = missing) -> int8, 0 = NA."""
    a = np.asarray(data)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3:
        raise ValueError("data must be an n x m x horizon array")
    if a.dtype.kind == "f":
        a = np.where(np.isnan(a), 0, a)
    a = np.where(a < 0, 0, a)
    return np.asfortranarray(a.astype(np.int8))


def _prior_rows(a, rows, cols, fill, name):
    """Accept the legacy 2-row form of a 3-row prior matrix (gpirt-senate.R:75)."""
    if a is None:
        return None
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != cols or a.shape[0] not in (rows, rows - 1):
        raise ValueError("%s must be %d x %d (or %d x %d), got %r" % (name, rows, cols, rows - 1, cols, a.shape))
    if a.shape[0] == rows - 1:
        a = np.vstack([a, np.full((1, cols), fill)])
    return a


def gpirtMCMC(data, sample_iterations, burn_iterations, THIN=1, CHAIN=1,
              beta_prior_means=None, beta_prior_sds=None, theta_prior_means=None,
              theta_prior_sds=None, theta_os=1.0, theta_ls=10.0, theta_init=None,
              vote_codes=None, thresholds=None, SEED=1, constant_IRF=0, KERNEL="RBF",
              beta_proposal_sds=None, *, return_f=True, return_fstar=True, theta_mode="ess",
              jitter=1e-6, max_ess_rounds=200, item_range=None, rank=0, world=1, allreduce=None,
              progress=None, summary=False, summary_quadratic=False, summary_drop_first_theta=False, flags=0):
    """Run ``CHAIN`` independent chains; returns a list with one dict per chain holding

    ``theta [K, n, H]``, ``f [K, n, m, H]``, ``fstar [K, 1001, m, H]``, ``beta [K, 3, m, H]``,
    ``threshold [K, m, C+1, H]`` (K = sample_iterations // THIN), so that
    ``samples["fstar"][it][:, j, h]`` reads like the R list ``samples$fstar[[it]][, j, h]``.
    ``f``/``fstar`` exclude the linear part ``beta[0] + beta[1]*theta`` as in the reference
    (gpirt-sythetic.R:153-154).  ``vote_codes`` and ``beta_proposal_sds`` are accepted and
    ignored (legacy arguments, gpirt_CIRI.R:135).  ``summary=True`` adds ``"summary"``: the
    posterior summaries accumulated on the device (``gpirt_amd.post`` turns them into the
    reference's evaluation metrics), so that ``return_f`` / ``return_fstar`` can be False;
    ``summary_drop_first_theta`` leaves the first kept draw out of its theta mean / sd, as the
    reference's ``samples$theta[-1, i, h]`` does (gpirt-sythetic.R:123).
    ``flags`` is OR-ed into ``gpirt_config.flags`` (e.g. ``_abi.GENERATOR_PRIOR``: the prior
    factor in generator form, see ``include/gpirt.h``).
    """
    del vote_codes, beta_proposal_sds
    y = responses_from_r(data)
    n, m, H = y.shape
    Ccat = int(y.max())
    if Ccat < 2:
        Ccat = 2
    if thresholds is not None:
        thresholds = np.asarray(thresholds, dtype=np.float64)
        Ccat = max(Ccat, thresholds.shape[0] - 1)
    kernels = {"RBF": _abi.KERNEL_RBF, "Matern": _abi.KERNEL_MATERN52, "MATERN": _abi.KERNEL_MATERN52}
    if KERNEL not in kernels:
        raise ValueError("KERNEL must be 'RBF' or 'Matern'")
    modes = {"ess": _abi.THETA_ESS, "grid": _abi.THETA_GRID}
    flags = int(flags) | (_abi.RETURN_F if return_f else 0) | (_abi.RETURN_FSTAR if return_fstar else 0)
    bpm = _prior_rows(beta_prior_means, 3, m, 0.0, "beta_prior_means")
    bps = _prior_rows(beta_prior_sds, 3, m, 0.0, "beta_prior_sds")
    thin = max(int(THIN), 1)
    K = int(sample_iterations) // thin
    total = int(sample_iterations) + int(burn_iterations)
    chains = []
    keep = []  # callbacks must outlive the call
    for chain in range(int(CHAIN)):
        cfg = _abi.default_config(
            n=n, m=m, H=H, C=Ccat, sample_iterations=int(sample_iterations),
            burn_iterations=int(burn_iterations), thin=thin, chain=chain, seed=int(SEED),
            theta_os=float(theta_os), theta_ls=float(theta_ls), kernel=kernels[KERNEL],
            constant_irf=int(constant_IRF), jitter=float(jitter), theta_mode=modes[theta_mode],
            max_ess_rounds=int(max_ess_rounds), flags=flags, rank=int(rank), world=int(world))
        if item_range is not None:
            cfg.item_begin, cfg.item_end = int(item_range[0]), int(item_range[1])
        if allreduce is not None:
            cb = _abi.ALLREDUCE_FN(lambda ptr, count, user, _f=allreduce: int(_f(ptr, count) or 0))
            keep.append(cb)
            cfg.allreduce = cb
        if progress is not None:
            pcb = _abi.PROGRESS_FN(lambda it, tot, user, _f=progress: int(_f(it, tot) or 0))
            keep.append(pcb)
            cfg.progress = pcb
        inp = HostInputs(cfg, y, theta_init=theta_init, beta_prior_means=bpm, beta_prior_sds=bps,
                         theta_prior_means=theta_prior_means, theta_prior_sds=theta_prior_sds,
                         thresholds=thresholds)
        mloc, P = local_shape(cfg)
        out = _abi.GpirtOutputs()
        th = np.empty((n, H, K), order="F")
        be = np.empty((3, mloc, P, K), order="F")
        tr = np.empty((mloc, Ccat + 1, P, K), order="F")
        ll = np.empty(total)
        ff = np.empty((n, mloc, H, K), order="F") if return_f else None
        fs = np.empty((cfg.n_grid, mloc, P, K), order="F") if return_fstar else None
        out.theta = th.ctypes.data_as(_dp)
        out.beta = be.ctypes.data_as(_dp)
        out.threshold = tr.ctypes.data_as(_dp)
        out.loglik = ll.ctypes.data_as(_dp)
        if ff is not None:
            out.f = ff.ctypes.data_as(_dp)
        if fs is not None:
            out.fstar = fs.ctypes.data_as(_dp)
        summ = None
        if summary:
            from .sampler import fetch_summary
            summ = fetch_summary(cfg, lambda sm: check(lib().gpirt_run_summary(
                C.byref(cfg), *inp.args(), C.byref(out), C.byref(sm),
                (_abi.SUMMARY_QUADRATIC if summary_quadratic else 0) |
                (_abi.SUMMARY_DROP_FIRST_THETA if summary_drop_first_theta else 0))))
        else:
            check(lib().gpirt_run(C.byref(cfg), *inp.args(), C.byref(out)))

        def per_period(a):  # [.., P, K] -> [K, .., H]
            a = np.moveaxis(a, -1, 0)
            if P != H:
                a = np.broadcast_to(a, a.shape[:-1] + (H,))
            return a

        chains.append(dict(
            theta=np.moveaxis(th, -1, 0),
            f=None if ff is None else np.moveaxis(ff, -1, 0),
            fstar=None if fs is None else per_period(fs),
            beta=per_period(be),
            threshold=per_period(tr),
            loglik=ll,
            summary=summ,
            diagnostics=dict(ess_f_rounds=out.ess_f_rounds, ess_param_rounds=out.ess_param_rounds,
                             ess_theta_rounds=out.ess_theta_rounds, ess_capped=out.ess_capped,
                             jitter_retries=out.jitter_retries, seconds=out.seconds_total),
        ))
    return chains
