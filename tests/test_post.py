"""The post-processing restatement (gpirt_amd/post.py) against a literal, loop-by-loop
transcription of the reference's R code on a tiny case, and the reference's acceptance
metrics on its own fixture with a short CPU-oracle chain."""
import numpy as np
from scipy.stats import norm

from conftest import load_fixture
from gpirt_amd import _abi, post


def _fake_draws(K=5, n=7, m=3, H=2, C=4, G=1001, seed=0):
    rng = np.random.default_rng(seed)
    thr = np.sort(rng.normal(size=(K, m, C - 1, H)), axis=2)
    inf = np.full((K, m, 1, H), np.inf)
    return dict(theta=rng.normal(size=(K, n, H)),
                fstar=0.3 * rng.normal(size=(K, G, m, H)).cumsum(axis=1) / 30,
                beta=rng.normal(size=(K, 3, m, H)),
                threshold=np.concatenate([-inf, thr, inf], axis=2))


def test_summarize_draws_matches_the_r_loops():
    s = _fake_draws()
    K, G, m, H = s["fstar"].shape
    C = s["threshold"].shape[2] - 1
    xs = post.grid()
    out = post.summarize_draws(s)
    # gpirt-sythetic.R:122-123
    assert np.allclose(out["theta_mean"], s["theta"].mean(0))
    assert np.allclose(out["theta_sd"], s["theta"].std(0, ddof=1))
    # :151-156 (linear part), :159-167 (ordinal_lls), getprob_gpirt.R:6-14, on a few grid points
    for g in (0, 311, 500, 1000):
        for j in range(m):
            for h in range(H):
                ps = np.zeros((K, C))
                am = np.zeros(K)
                for it in range(K):
                    f = s["fstar"][it][g, j, h] + s["beta"][it][0, j, h] + s["beta"][it][1, j, h] * xs[g]
                    b = s["threshold"][it][j, :, h]
                    for c in range(C):
                        ps[it, c] = norm.cdf(b[c + 1] - f) - norm.cdf(b[c] - f)
                    am[it] = 1 + int(np.argmax(np.log(ps[it])))
                assert np.allclose(out["prob_mean"][g, j, h], ps.mean(0), atol=1e-14)
                assert abs(out["argmax_mean"][g, j, h] - am.mean()) < 1e-14


def test_evaluate_follows_the_r_script_on_a_tiny_case():
    s = _fake_draws(K=6, n=9, m=2, H=2, C=3, seed=3)
    rng = np.random.default_rng(1)
    n, m, H = 9, 2, 2
    data = rng.integers(0, 4, size=(n, m, H))          # 0 = NA
    train = rng.integers(0, 2, size=(n, m, H))
    theta_true = rng.normal(size=(n, H))
    truth = dict(alpha=rng.normal(size=(m, H)), beta=rng.normal(size=(m, H)),
                 thresholds=np.array([-np.inf, -0.4, 0.5, np.inf]))
    summ = post.summarize_draws(s)
    ev = post.evaluate(summ, data, train, theta_true, truth)
    # :131-133 centring, :190 index, :253-256 correlation
    pt = summ["theta_mean"] - summ["theta_mean"].mean(0)
    assert np.allclose(ev["pred_theta"], pt)
    for h in range(H):
        assert abs(ev["cor_theta"][h] - np.corrcoef(theta_true[:, h], pt[:, h])[0, 1]) < 1e-12
    lls, accs = {0: [], 1: []}, {0: [], 1: []}
    for i in range(n):
        for j in range(m):
            for h in range(H):
                if data[i, j, h] > 0:
                    idx = 1 + int((pt[i, h] + 5) * 100)
                    p = summ["prob_mean"][idx - 1, j, h]
                    ypred = round(float(summ["argmax_mean"][idx - 1, j, h]))   # python round: half-even, like R
                    lls[int(train[i, j, h])].append(np.log(p[data[i, j, h] - 1]))
                    accs[int(train[i, j, h])].append(ypred == data[i, j, h])
    assert abs(ev["train_ll"] - np.mean(lls[1])) < 1e-12 and abs(ev["pred_ll"] - np.mean(lls[0])) < 1e-12
    assert abs(ev["train_acc"] - np.mean(accs[1])) < 1e-12 and abs(ev["pred_acc"] - np.mean(accs[0])) < 1e-12
    # :230-251 with the 2PL truth of true_irf.R:1-4
    xs = post.grid()[300:701]
    order = np.arange(1, 4)
    for j in range(m):
        for h in range(H):
            irf = truth["alpha"][j, h] + truth["beta"][j, h] * xs
            cdf = norm.cdf(truth["thresholds"][None, :] - irf[:, None])
            t = ((cdf[:, 1:] - cdf[:, :-1]) * order).sum(1)
            if np.sign(ev["cor_theta"][h]) < 0:
                t = t[::-1]
            mine = (summ["prob_mean"][300:701, j, h] * order).sum(1)
            assert abs(ev["cor_icc"][j, h] - np.corrcoef(mine, t)[0, 1]) < 1e-10
            assert abs(ev["rmse_icc"][j, h] - np.sqrt(np.mean((mine - t) ** 2))) < 1e-12
    assert len(post.report(ev)) == 9


def test_reference_acceptance_metrics_on_its_fixture_with_the_cpu_chain():
    """gpirt-sythetic.R:275-283 on the Wiener fixture (3 periods, short chain): the chain must
    recover theta and the response curves of the generating model."""
    from oracle import oracle
    d = load_fixture("Wiener_C_5")
    H = 3
    y, ytr, tidx, theta = d["y"][:, :, :H], d["y_train"][:, :, :H], d["train_idx"][:, :, :H], d["theta"][:, :H]
    n, m, _ = y.shape
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=1, theta_os=1.0, theta_ls=H / 2,
                              sample_iterations=60, burn_iterations=60, thin=4,
                              flags=_abi.RETURN_FSTAR)
    rng = np.random.default_rng(1)
    r = oracle.run(cfg, ytr, theta_init=theta + 0.05 * rng.normal(size=theta.shape))
    samples = dict(theta=np.moveaxis(r["theta"], -1, 0), fstar=np.moveaxis(r["fstar"], -1, 0),
                   beta=np.moveaxis(r["beta"], -1, 0), threshold=np.moveaxis(r["threshold"], -1, 0))
    truth = dict(thresholds=d["thresholds"], anchor_xs=d["anchor_xs"][:, :, :H],
                 anchor_ys=d["anchor_ys"][:, :, :H], SIGMA=d["SIGMA"])
    ev = post.evaluate(post.summarize_draws(samples), y, tidx, theta, truth)
    assert ev["mean_abs_cor_theta"] > 0.9
    assert ev["mean_abs_cor_icc"] > 0.8
    assert ev["train_acc"] > 0.3 and ev["pred_acc"] > 0.3      # 5 categories: chance is 0.2
    assert ev["pred_cells"] > 0 and ev["train_cells"] > ev["pred_cells"]


def test_holdout_mask_and_per_period_scores():
    rng = np.random.default_rng(5)
    data = rng.integers(0, 4, size=(40, 3, 5))
    mask = post.holdout_mask(data, periods=(3, 4), drop_percent=20, seed=1)
    assert mask[:, :, :3].all()
    for h in (3, 4):
        for j in range(3):
            held = np.flatnonzero(mask[:, j, h] == 0)
            assert held.size == 8 and (data[held, j, h] > 0).all()
    s = _fake_draws(K=4, n=40, m=3, H=5, C=3, seed=7)
    summ = post.summarize_draws(s)
    ev = post.evaluate(summ, data, mask, prediction="argmax_mean")
    assert np.isnan(ev["pred_ll_by_period"][:3]).all() and np.isfinite(ev["pred_ll_by_period"][3:]).all()
    assert ev["pred_cells"] == 48
    w = np.array([(mask[:, :, h] == 0).sum() for h in (3, 4)])
    assert abs(ev["pred_ll"] - (ev["pred_ll_by_period"][3:] * w).sum() / w.sum()) < 1e-12
