"""Chain diagnostics the reference computes in R after a run: effective sample size
(``coda::effectiveSize``, gpirt-senate.R:130-137, gpirt_esm.R:66-84) and the potential scale
reduction factor across chains (``rstan::Rhat``, gpirt_esm.R:90-100).

Both are restated from the published algorithms of those packages (coda's ``spectrum0.ar``:
spectral density at zero from a Yule-Walker AR fit with AIC order selection; rstan's
rank-normalised split R-hat).  R is not available to compare numbers with: checked against
closed-form cases only (parity unpinned).
"""
import numpy as np
from scipy.special import ndtri
from scipy.stats import rankdata


def _ar_yule_walker(x):
    """R's ``ar(x, aic = TRUE, method = "yule-walker")``: returns (coefficients, var.pred)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    order_max = min(n - 1, int(np.floor(10 * np.log10(n))))
    xc = x - x.mean()
    r = np.array([np.dot(xc[:n - k], xc[k:]) / n for k in range(order_max + 1)])
    if r[0] == 0:
        return np.zeros(0), 0.0
    # Levinson-Durbin: coefficients and innovation variance for every order
    phi = np.zeros((order_max + 1, order_max + 1))
    v = np.zeros(order_max + 1)
    v[0] = r[0]
    for k in range(1, order_max + 1):
        a = (r[k] - np.dot(phi[k - 1, 1:k], r[1:k][::-1])) / v[k - 1]
        phi[k, k] = a
        phi[k, 1:k] = phi[k - 1, 1:k] - a * phi[k - 1, 1:k][::-1]
        v[k] = v[k - 1] * (1 - a * a)
    aic = n * np.log(v) + 2 * np.arange(order_max + 1) + 2
    order = int(np.argmin(aic))
    var_pred = v[order] * n / (n - (order + 1))
    return phi[order, 1:order + 1], var_pred


def effective_size(x):
    """``coda::effectiveSize`` of one series, or of every column of a 2-D array."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 2:
        return np.array([effective_size(x[:, j]) for j in range(x.shape[1])])
    n = x.size
    ar, var_pred = _ar_yule_walker(x)
    spec = var_pred / (1.0 - ar.sum()) ** 2
    return 0.0 if spec == 0 else n * x.var(ddof=1) / spec


def _split(sims):
    n = sims.shape[0]
    half = n // 2
    return np.concatenate([sims[:half], sims[n - half:]], axis=1)


def _z_scale(sims):
    r = rankdata(sims, method="average").reshape(sims.shape)
    return ndtri((r - 0.375) / (sims.size + 0.25))


def _rhat_basic(sims):
    n = sims.shape[0]
    means, variances = sims.mean(axis=0), sims.var(axis=0, ddof=1)
    between = n * means.var(ddof=1)
    within = variances.mean()
    return float(np.sqrt(((n - 1) / n * within + between / n) / within))


def rhat(sims):
    """``rstan::Rhat``: `sims` [draws, chains]; the larger of the rank-normalised split R-hat
    of the draws and of their absolute deviations from the median."""
    sims = np.asarray(sims, dtype=np.float64)
    if sims.ndim == 1:
        sims = sims[:, None]
    bulk = _rhat_basic(_z_scale(_split(sims)))
    tail = _rhat_basic(_z_scale(_split(np.abs(sims - np.median(sims)))))
    return max(bulk, tail)


def theta_diagnostics(chains):
    """ESS of theta per (respondent, period) from the first chain and, with several chains,
    R-hat across them: `chains` is what `gpirtMCMC` returns."""
    th = [np.asarray(c["theta"]) for c in chains]               # [K, n, H] each
    K, n, H = th[0].shape
    ess = np.array([[effective_size(th[0][:, i, h]) for h in range(H)] for i in range(n)])
    out = dict(ess=ess)
    if len(th) > 1:
        out["rhat"] = np.array([[rhat(np.stack([t[:, i, h] for t in th], axis=1)) for h in range(H)]
                                for i in range(n)])
    return out
