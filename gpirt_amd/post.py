"""Posterior summaries and the evaluation metrics the reference prints after a run
(gpirt-sythetic.R:111-283, getprob_gpirt.R:1-20, true_irf.R:1-15), as vectorised numpy.

Two sources feed the same `evaluate`:

* `summarize_draws(samples)` -- from the kept draws a `gpirtMCMC` call returned (what the R
  script does, draw by draw);
* the device-side running sums (`gpirtMCMC(..., summary=True)["summary"]`,
  `Sampler.accumulate()/summary()`), for runs whose f / f* draws never leave the GPU.

Indices follow the reference: the grid is xs = seq(-5, 5, 0.01); a respondent's predictive
index is ``1 + as.integer((theta + 5) * 100)`` (truncation, gpirt-sythetic.R:190); item
characteristic curves are compared on xs[301:701] (:231).
"""
import numpy as np
from scipy.special import ndtr

GRID_LO, GRID_STEP, N_GRID = -5.0, 0.01, 1001
ICC_SLICE = slice(300, 701)     # xs[301:701] in R's 1-based, inclusive indexing


def grid(n_grid=N_GRID, lo=GRID_LO, step=GRID_STEP):
    return lo + step * np.arange(n_grid)


def category_probs(F, thresholds):
    """P(y = c) = Phi(b_c - F) - Phi(b_{c-1} - F) (getprob_gpirt.R:9-12): F [...],
    thresholds [..., C+1] broadcastable against F[..., None]; returns [..., C]."""
    z = thresholds - np.asarray(F)[..., None]
    cdf = ndtr(z)
    return cdf[..., 1:] - cdf[..., :-1]


def summarize_draws(samples, quadratic=False, xs=None, drop_first_theta=False):
    """The quantities of `gpirt_summary` from kept draws (a chain dict of `gpirtMCMC`):
    ``theta_mean/theta_sd [n,H]`` (:122-123, sd with n-1), ``prob_mean [G,m,H,C]`` and
    ``argmax_mean [G,m,H]`` with F* = fstar + beta_0 + beta_1 xs (:153-154; `quadratic`
    adds beta_2 xs^2).  The reference takes theta's mean and sd over ``samples$theta[-1, i, h]``,
    i.e. without the first kept draw (:123, :131-132): `drop_first_theta` does the same; the
    probability sums always run over all draws, as the reference's do (:159-214)."""
    theta = np.asarray(samples["theta"])                  # [K, n, H]
    if drop_first_theta and theta.shape[0] > 1:
        theta = theta[1:]
    fstar = np.asarray(samples["fstar"])                  # [K, G, m, H]
    beta = np.asarray(samples["beta"])                    # [K, 3, m, H]
    thr = np.asarray(samples["threshold"])                # [K, m, C+1, H]
    K, G = fstar.shape[0], fstar.shape[1]
    xs = grid(G) if xs is None else np.asarray(xs)
    C = thr.shape[2] - 1
    prob = np.zeros(fstar.shape[1:] + (C,))
    arg = np.zeros(fstar.shape[1:])
    for k in range(K):
        F = fstar[k] + beta[k, 0][None] + beta[k, 1][None] * xs[:, None, None]
        if quadratic:
            F = F + beta[k, 2][None] * (xs * xs)[:, None, None]
        p = category_probs(F, np.moveaxis(thr[k], 1, -1)[None])     # [G, m, H, C]
        prob += p
        arg += 1 + np.argmax(p, axis=-1)                  # which.max: first maximum
    return dict(theta_mean=theta.mean(axis=0),
                theta_sd=theta.std(axis=0, ddof=1) if theta.shape[0] > 1 else np.full(theta.shape[1:], np.nan),
                prob_mean=prob / K, argmax_mean=arg / K, draws=K)


def _per_period(a, H):
    """[.., P, ..] summaries of a constant-IRF run (P = 1) read for every period."""
    return a if a.shape[2] == H else np.broadcast_to(a, a.shape[:2] + (H,) + a.shape[3:])


def true_irf(truth, j, h, x):
    """The generating response function of item j, period h at x (true_irf.R:1-15):
    2PL ``alpha + beta x``, otherwise the GP interpolant through the anchors."""
    if "alpha" in truth:
        return truth["alpha"][j, h] + truth["beta"][j, h] * x
    ax, ay = truth["anchor_xs"][j, :, h], truth["anchor_ys"][j, :, h]
    sig = float(np.ravel(truth["SIGMA"])[0])
    Kaa = np.exp(-0.5 * (ax[:, None] - ax[None, :]) ** 2 / sig ** 2) + 1e-6 * np.eye(ax.size)
    K1 = np.exp(-0.5 * (x[:, None] - ax[None, :]) ** 2 / sig ** 2)
    return K1 @ (np.linalg.pinv(Kaa) @ ay)


def _cor(a, b):
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def holdout_mask(data, periods, drop_percent, seed=0):
    """train_idx for the k-step-ahead experiments (TAPS/gpirt_TAPS_holdout.R:50-59): in each of
    `periods` (0-based) and for each item, ``int(drop_percent/100 * n)`` of the answered cells
    are held out.  Returns int8 [n, m, H], 1 = train, 0 = held out; pass
    ``np.where(mask == 1, data, 0)`` to the sampler.  (R's own ``sample()`` stream cannot be
    reproduced here, only its rule.)"""
    data = np.asarray(data)
    n, m, H = data.shape
    rng = np.random.default_rng(seed)
    mask = np.ones((n, m, H), dtype=np.int8)
    for h in periods:
        for j in range(m):
            answered = np.flatnonzero(data[:, j, h] > 0)
            k = min(int(drop_percent / 100.0 * n), answered.size)
            if k > 0:
                mask[rng.choice(answered, size=k, replace=False), j, h] = 0
    return mask


def evaluate(summary, data, train_idx=None, theta_true=None, truth=None, prediction="mean_argmax"):
    """The reference's evaluation (gpirt-sythetic.R:131-143,187-283) from a summary.

    data       int [n, m, H], 1..C, 0 = NA (the FULL data: held-out cells included)
    train_idx  [n, m, H], 1 = used for training, 0 = held out (None: everything trained)
    theta_true [n, H] generating abilities (None skips the theta metrics)
    truth      dict with ``thresholds`` and either ``alpha``/``beta`` (2PL) or
               ``anchor_xs``/``anchor_ys``/``SIGMA`` (None skips the ICC metrics)
    prediction "mean_argmax": round(mean over draws of which.max), gpirt-sythetic.R:197,208;
               "argmax_mean": which.max of the posterior-mean probabilities, the rule of the
               hold-out scripts (TAPS/gpirt_TAPS_holdout.R:176-180)

    Predictive log-likelihood of a cell: log of the posterior-mean probability of the observed
    category at the respondent's grid index -- the reference's commented-out
    ``log(colMeans(exp(lls)))`` (:200-204).  Its live line ``ll = log(exp(lls))`` leaves a
    draws x C matrix that ``ll[data]`` then indexes linearly (draw number y of category 1);
    that accident is not reproduced.
    """
    data = np.asarray(data)
    n, m, H = data.shape
    pm = _per_period(np.asarray(summary["prob_mean"]), H)         # [G, m, H, C]
    am = _per_period(np.asarray(summary["argmax_mean"]), H)       # [G, m, H]
    G = pm.shape[0]
    xs = grid(G)
    pred_theta = summary["theta_mean"] - summary["theta_mean"].mean(axis=0, keepdims=True)  # :131-133
    sd = np.asarray(summary["theta_sd"])
    out = dict(pred_theta=pred_theta, pred_theta_sd=sd, mean_theta_sd=float(np.mean(sd)))

    sign = np.ones(H)
    if theta_true is not None:
        theta_true = np.asarray(theta_true)
        cor_theta = np.array([_cor(theta_true[:, h], pred_theta[:, h]) for h in range(H)])
        sign = np.sign(cor_theta)
        with np.errstate(divide="ignore", invalid="ignore"):
            z = (theta_true * sign[None] - pred_theta) / sd
            ll = -0.5 * z * z - np.log(sd) - 0.5 * np.log(2 * np.pi)                          # :137-142
        out.update(cor_theta=cor_theta, mean_abs_cor_theta=float(np.mean(np.abs(cor_theta))),
                   mean_theta_ll=float(np.mean(ll[np.isfinite(ll)])))

    # predictive / training log-likelihood and accuracy (:187-227)
    idx = (1 + ((pred_theta + 5.0) * 100.0).astype(np.int64)) - 1                            # :190, 0-based
    idx = np.clip(idx, 0, G - 1)
    obs = data > 0
    ii, jj, hh = np.nonzero(obs)
    yy = data[ii, jj, hh].astype(np.int64)
    gi = idx[ii, hh]
    with np.errstate(divide="ignore"):
        ll_cell = np.log(pm[gi, jj, hh, yy - 1])
    if prediction == "mean_argmax":
        y_pred = np.rint(am[gi, jj, hh])                                                      # :208
    else:   # "argmax_mean": which.max of the posterior-mean probabilities (gpirt_TAPS_holdout.R:179)
        y_pred = 1 + np.argmax(pm[gi, jj, hh], axis=-1)
    hit = (y_pred == yy)
    tr = np.ones(ii.size, dtype=bool) if train_idx is None else (np.asarray(train_idx)[ii, jj, hh] != 0)
    for name, mask in (("train", tr), ("pred", ~tr)):
        fin = mask & np.isfinite(ll_cell)
        out[name + "_ll"] = float(ll_cell[fin].mean()) if fin.any() else float("nan")
        out[name + "_acc"] = float(hit[fin].mean()) if fin.any() else float("nan")
        out[name + "_cells"] = int(mask.sum())
        # per period: the k-step-ahead tables of TAPS/gpirt_TAPS_holdout.R:146-199
        out[name + "_ll_by_period"] = np.array(
            [ll_cell[fin & (hh == h)].mean() if (fin & (hh == h)).any() else np.nan for h in range(H)])
        out[name + "_acc_by_period"] = np.array(
            [hit[fin & (hh == h)].mean() if (fin & (hh == h)).any() else np.nan for h in range(H)])

    # item characteristic curves (:230-251): expected category on xs[301:701]
    if truth is not None:
        C = pm.shape[-1]
        order = np.arange(1, C + 1)
        xi = xs[ICC_SLICE]
        icc = (pm[ICC_SLICE] * order).sum(axis=-1)                                            # [401, m, H]
        thr = np.asarray(truth["thresholds"], dtype=np.float64)
        cor_icc, rmse_icc = np.zeros((m, H)), np.zeros((m, H))
        true_icc = np.zeros_like(icc)
        for h in range(H):
            for j in range(m):
                t = (category_probs(true_irf(truth, j, h, xi), thr[None]) * order).sum(axis=-1)
                # getprobs_gpirt labels the curve with sign * xs and the summary is grouped (hence
                # sorted) by that label: a negative sign reverses the curve (:239-243)
                true_icc[:, j, h] = t[::-1] if sign[h] < 0 else t
                cor_icc[j, h] = _cor(icc[:, j, h], true_icc[:, j, h])
                rmse_icc[j, h] = float(np.sqrt(np.mean((icc[:, j, h] - true_icc[:, j, h]) ** 2)))
        out.update(icc=icc, true_icc=true_icc, cor_icc=cor_icc, rmse_icc=rmse_icc,
                   mean_abs_cor_icc=float(np.mean(np.abs(cor_icc))), mean_rmse_icc=float(np.mean(rmse_icc)))
    return out


def report(ev):
    """The nine numbers the reference prints (gpirt-sythetic.R:275-283), in its order."""
    keys = ("mean_abs_cor_theta", "mean_theta_sd", "mean_theta_ll", "train_ll", "train_acc",
            "pred_ll", "pred_acc", "mean_abs_cor_icc", "mean_rmse_icc")
    return [ev.get(k, float("nan")) for k in keys]
