"""Synthetic ordinal-probit GPIRT data, restating the "GP"/"Wiener" recipe of the
reference's generator (generate_data.R:36-48,92-119,121-209) with numpy's generator in
place of R's Mersenne-Twister stream: the distribution matches, the values do not."""
import numpy as np
from scipy.special import ndtr


def se_kernel(a, b, sigma=1.0):
    """exp(-0.5 d^2 / sigma^2)  (getprob_gpirt.R:41-51)."""
    d = np.subtract.outer(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
    return np.exp(-0.5 * d * d / (sigma * sigma))


def make_thresholds(C, rng):
    """generate_data.R:36-48: C-1 points on [-2,2], jittered, centred, scaled to sd 1."""
    b = np.empty(C + 1)
    b[0], b[C] = -np.inf, np.inf
    if C > 2:
        t = np.linspace(-2, 2, C - 1) + rng.normal(0, 0.25, C - 1)
        t = t - t.mean()
        t = t / t.std(ddof=1)
        b[1:C] = np.sort(t)
    else:
        b[1] = rng.normal(0, 0.25)
    return b


def generate(n, m, C, H=1, seed=1, kind="GP", num_anchor=50, sigma=1.0, item_chunk=512):
    """Returns dict(y int8 [n,m,H] (no missing cells), theta [n,H], thresholds [C+1],
    anchor_xs [num_anchor], anchor_ys [m,num_anchor,H])."""
    rng = np.random.default_rng(seed)
    thresholds = make_thresholds(C, rng)
    theta = np.zeros((n, H))
    if kind == "Wiener":                                    # generate_data.R:170-178
        theta[:, 0] = rng.normal(size=n)
        for h in range(1, H):
            theta[:, h] = theta[:, h - 1] + rng.normal(0, 0.1, n)
    elif H > 1:                                             # :124-131
        Kt = se_kernel(np.arange(1, H + 1), np.arange(1, H + 1), int(1 + H / 2)) + 1e-6 * np.eye(H)
        theta = rng.normal(size=(n, H)) @ np.linalg.cholesky(Kt).T
    else:
        theta[:, 0] = rng.normal(size=n)
    ax = np.linspace(-2, 2, num_anchor)                     # :145
    Ka = se_kernel(ax, ax, sigma) + 1e-6 * np.eye(num_anchor)
    La = np.linalg.cholesky(Ka)
    Kinv = np.linalg.pinv(Ka)                               # ginv(K), :104
    y = np.empty((n, m, H), dtype=np.int8, order="F")
    anchor_ys = np.empty((m, num_anchor, H))
    for h in range(H):
        ay = rng.normal(size=(m, num_anchor)) @ La.T        # :148
        slope = (2 * rng.integers(0, 2, m) - 1) * rng.normal(size=m)   # :150
        ay = ay + slope[:, None] * ax[None, :]              # :154
        anchor_ys[:, :, h] = ay
        K1 = se_kernel(theta[:, h], ax, sigma)              # dnorm ratio of :106
        A = K1 @ Kinv
        for j0 in range(0, m, item_chunk):
            j1 = min(m, j0 + item_chunk)
            f = A @ ay[j0:j1].T                             # [n, chunk]  (:107)
            cdf = ndtr(thresholds[None, None, 1:C] - f[:, :, None])   # P(y <= c), c = 1..C-1
            u = rng.uniform(size=f.shape)
            y[:, j0:j1, h] = 1 + (u[:, :, None] > cdf).sum(axis=2)    # inverse-cdf draw (:108-114)
    return dict(y=y, theta=theta, thresholds=thresholds, anchor_xs=ax, anchor_ys=anchor_ys)
