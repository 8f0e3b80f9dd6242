"""Resident sampler: the chain's state stays in HBM between calls (``gpirt_sampler_*``)."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib, check, _dp, _bp

PHASES = ("k_build", "potrf", "nu", "ess_f", "ess_param", "fstar", "theta", "snap", "normals")


def _f(a):
    return None if a is None else np.asfortranarray(np.asarray(a, dtype=np.float64))


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def local_shape(cfg):
    """(m_loc, P): items owned by this rank and number of IRF groups."""
    j0, j1 = (cfg.item_begin, cfg.item_end) if cfg.item_end > cfg.item_begin else (0, cfg.m)
    return j1 - j0, (1 if cfg.constant_irf else cfg.H)


class HostInputs:
    """Caller-side buffers in the ABI's layouts, kept alive across a call."""

    def __init__(self, cfg, y, theta_init=None, beta_prior_means=None, beta_prior_sds=None,
                 theta_prior_means=None, theta_prior_sds=None, thresholds=None):
        y = np.asarray(y)
        if y.shape != (cfg.n, cfg.m, cfg.H) and y.size == cfg.n * cfg.m * cfg.H:
            y = y.reshape((cfg.n, cfg.m, cfg.H), order="F")
        if y.shape != (cfg.n, cfg.m, cfg.H):
            raise ValueError("data must be [n, m, H] = %r, got %r" % ((cfg.n, cfg.m, cfg.H), y.shape))
        self.y = np.asfortranarray(y.astype(np.int8))
        self.theta_init = _f(theta_init)
        self.bpm, self.bps = _f(beta_prior_means), _f(beta_prior_sds)
        self.tpm, self.tps = _f(theta_prior_means), _f(theta_prior_sds)
        self.thr = _f(thresholds)
        for a, shp, nm in ((self.theta_init, (cfg.n, cfg.H), "theta_init"),
                           (self.bpm, (3, cfg.m), "beta_prior_means"),
                           (self.bps, (3, cfg.m), "beta_prior_sds"),
                           (self.tpm, (2, cfg.n), "theta_prior_means"),
                           (self.tps, (2, cfg.n), "theta_prior_sds"),
                           (self.thr, (cfg.C + 1,), "thresholds")):
            if a is not None and a.shape != shp:
                raise ValueError("%s must have shape %r, got %r" % (nm, shp, a.shape))
        if self.thr is not None:
            self.thr = np.where(np.isfinite(self.thr), self.thr, 0.0)

    def args(self):
        return (self.y.ctypes.data_as(_bp), _d(self.theta_init), _d(self.bpm), _d(self.bps),
                _d(self.tpm), _d(self.tps), _d(self.thr))


def fetch_summary(cfg, call):
    """Allocate the buffers of a ``gpirt_summary``, let `call(summary)` fill them, return a dict."""
    mloc, P = local_shape(cfg)
    sm = _abi.GpirtSummary()
    sm.struct_size = C.sizeof(_abi.GpirtSummary)
    tm = np.empty((cfg.n, cfg.H), order="F")
    ts = np.empty((cfg.n, cfg.H), order="F")
    pm = np.empty((cfg.n_grid, mloc, P, cfg.C), order="F")
    am = np.empty((cfg.n_grid, mloc, P), order="F")
    sm.theta_mean, sm.theta_sd, sm.prob_mean, sm.argmax_mean = _d(tm), _d(ts), _d(pm), _d(am)
    call(sm)
    return dict(theta_mean=tm, theta_sd=ts, prob_mean=pm, argmax_mean=am, draws=int(sm.draws))


class Sampler:
    def __init__(self, cfg, y, **inputs):
        self.cfg = cfg
        self._inputs = HostInputs(cfg, y, **inputs)
        self._h = C.c_void_p()
        check(lib().gpirt_sampler_create(C.byref(cfg), *self._inputs.args(), C.byref(self._h)))

    def sweep(self, k=1):
        check(lib().gpirt_sampler_sweep(self._h, int(k)))

    def state(self, f=True, fstar=True):
        c = self.cfg
        mloc, P = local_shape(c)
        th = np.empty((c.n, c.H), order="F")
        ff = np.empty((c.n, mloc, c.H), order="F") if f else None
        fs = np.empty((c.n_grid, mloc, P), order="F") if fstar else None
        be = np.empty((3, mloc, P), order="F")
        tr = np.empty((mloc, c.C + 1, P), order="F")
        check(lib().gpirt_sampler_get_state(self._h, _d(th), _d(ff), _d(fs), _d(be), _d(tr)))
        return dict(theta=th, f=ff, fstar=fs, beta=be, threshold=tr)

    def phase_times(self):
        ms = np.zeros(len(PHASES))
        cnt = np.zeros(len(PHASES), dtype=np.int64)
        check(lib().gpirt_sampler_phase_times(self._h, _d(ms), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        return {p: (float(ms[i]), int(cnt[i])) for i, p in enumerate(PHASES)}

    def update_times(self):
        """(ms, launches, flop) of the factorisation's rank-512 trailing updates since creation."""
        ms, fl, cnt = C.c_double(), C.c_double(), C.c_int64()
        check(lib().gpirt_sampler_update_times(self._h, C.byref(ms), C.byref(cnt), C.byref(fl)))
        return float(ms.value), int(cnt.value), float(fl.value)

    def diagnostics(self):
        o = _abi.GpirtOutputs()
        check(lib().gpirt_sampler_diagnostics(self._h, C.byref(o)))
        return dict(ess_f_rounds=o.ess_f_rounds, ess_param_rounds=o.ess_param_rounds,
                    ess_theta_rounds=o.ess_theta_rounds, ess_capped=o.ess_capped,
                    jitter_retries=o.jitter_retries)

    def nu_rows_beside_factor(self):
        """Rows of nu = L Z that the last sweep computed on the second stream beside the
        factorisation (0: one launch after it)."""
        o = _abi.GpirtOutputs()
        check(lib().gpirt_sampler_diagnostics(self._h, C.byref(o)))
        return int(o.nu_rows_beside_factor)

    def nu_tail_panels(self):
        """512-column panels of L whose share of the rows after `nu_rows_beside_factor()` was added
        beside the last sweep's factorisation as well."""
        o = _abi.GpirtOutputs()
        check(lib().gpirt_sampler_diagnostics(self._h, C.byref(o)))
        return int(o.nu_tail_panels)

    def slice_second_share(self):
        """Units whose slice updates the last sweep ran as a second share on a stream of their own
        (0: one launch per step)."""
        o = _abi.GpirtOutputs()
        check(lib().gpirt_sampler_diagnostics(self._h, C.byref(o)))
        return int(o.slice_second_share)

    def factor_lookahead(self):
        """True if the last sweep's factorisation ran its trailing updates on the second stream,
        beside the next block columns' steps."""
        o = _abi.GpirtOutputs()
        check(lib().gpirt_sampler_diagnostics(self._h, C.byref(o)))
        return bool(o.factor_lookahead)

    def ll_checksum(self):
        """GPIRT_FIXED_ORDER_SUM only: 64-bit checksum of every log-likelihood sum that the
        respondent step's decisions have read so far (equal for chains whose sums agree bit for bit)."""
        v = C.c_uint64()
        check(lib().gpirt_sampler_ll_checksum(self._h, C.byref(v)))
        return int(v.value)

    def accumulate(self, quadratic=False, skip_theta=False):
        """Add the current state to the device-side running sums of the posterior summaries
        (`skip_theta`: leave theta out of this draw's sums)."""
        check(lib().gpirt_sampler_accumulate(
            self._h, (_abi.SUMMARY_QUADRATIC if quadratic else 0) | (_abi.SUMMARY_SKIP_THETA if skip_theta else 0)))

    def summary(self):
        """``theta_mean/theta_sd [n,H]``, ``prob_mean [n_grid,m_loc,P,C]``,
        ``argmax_mean [n_grid,m_loc,P]`` over the draws accumulated so far."""
        return fetch_summary(self.cfg, lambda sm: check(lib().gpirt_sampler_summary(self._h, C.byref(sm))))

    def checkpoint(self):
        """The chain's state as bytes; `restore` into a sampler built from the same
        configuration and data continues the chain bit for bit."""
        size = int(lib().gpirt_sampler_checkpoint_bytes(self._h))
        buf = np.empty(size, dtype=np.uint8)
        check(lib().gpirt_sampler_checkpoint(self._h, buf.ctypes.data_as(C.c_void_p), size))
        return buf.tobytes()

    def restore(self, blob):
        buf = np.frombuffer(blob, dtype=np.uint8)
        check(lib().gpirt_sampler_restore(self._h, buf.ctypes.data_as(C.c_void_p), int(buf.size)))

    def stream(self):
        return lib().gpirt_sampler_stream(self._h)

    def close(self):
        if self._h:
            lib().gpirt_sampler_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
