"""Chain-level parity at the reference's own settings (gpirt-sythetic.R:60-94): for each of the
reference's synthetic data files, 500 burn-in + 500 sampling sweeps, THIN 4, one chain,
`theta_init = theta + 0.05 N(0,1)` (:86), `theta_ls = horizon / 2` (:64), unit beta priors
(:77-78), the FULL response array as data (:88) -- once through the device library
(`gpirtMCMC` mirror) and once through the CPU oracle -- then the nine numbers the script prints
(:275-283) from `gpirt_amd.post.evaluate`: theta correlation, posterior sd, theta log-density,
train / held-out log-likelihood and accuracy, ICC correlation and RMSE against true_irf.R:1-15
on xs[301:701].

The two chains share the random streams, so they coincide until the first slice decision that
falls on different sides in the two arithmetics (f* carries ~1e-6 of solve error); from there
they are two realisations of the same Markov chain.  The Monte-Carlo error of every metric is
therefore measured, not assumed: the device chain is repeated with `N_SEEDS` further sampler
seeds and the spread of each metric over seeds is the yardstick the CPU-vs-device difference is
held to (`tolerance`).

Run as a script on a GPU box to regenerate the committed table:
    python tests/tier2_parity.py profiles/r03_tier2_parity.json
TEST INFRASTRUCTURE (it drives the CPU oracle): used by tests/test_gpu_tier2.py only.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

FIXTURES = (("2PL_C_2", 2), ("2PL_C_5", 5), ("Wiener_C_5", 5))
SAMPLE_ITERS, BURNOUT_ITERS, THIN = 500, 500, 4          # gpirt-sythetic.R:60-61,75
N_SEEDS = 4                                              # extra device chains that size the MC error
METRICS = ("mean_abs_cor_theta", "mean_theta_sd", "mean_theta_ll", "train_ll", "train_acc",
           "pred_ll", "pred_acc", "mean_abs_cor_icc", "mean_rmse_icc")
# |cpu - device| <= max(K_SIGMA * sd over seeds, floor): the floor covers metrics whose spread
# over a handful of seeds happens to come out tiny
K_SIGMA = 5.0
FLOORS = dict(mean_abs_cor_theta=0.01, mean_theta_sd=0.01, mean_theta_ll=0.15, train_ll=0.01,
              train_acc=0.01, pred_ll=0.02, pred_acc=0.015, mean_abs_cor_icc=0.02, mean_rmse_icc=0.03)


def load(stem):
    from conftest import load_fixture
    d = load_fixture(stem)
    truth = {k: d[k] for k in d.files if k in ("thresholds", "alpha", "beta", "anchor_xs", "anchor_ys", "SIGMA")}
    return d["y"], d["train_idx"], d["theta"], truth


def settings(y, theta):
    n, m, H = y.shape
    init = theta + 0.05 * np.random.default_rng(1).normal(size=theta.shape)     # :86
    return dict(theta_os=1.0, theta_ls=H / 2.0, theta_init=init, bpm=np.zeros((3, m)), bps=np.ones((3, m)))


def device_chain(y, C, st, seed):
    from gpirt_amd.mcmc import gpirtMCMC
    t0 = time.perf_counter()
    ch = gpirtMCMC(y, SAMPLE_ITERS, BURNOUT_ITERS, THIN=THIN, CHAIN=1, beta_prior_means=st["bpm"],
                   beta_prior_sds=st["bps"], theta_os=st["theta_os"], theta_ls=st["theta_ls"],
                   theta_init=st["theta_init"], KERNEL="RBF", thresholds=None, SEED=seed,
                   constant_IRF=0, return_f=False, return_fstar=True)[0]
    return ch, time.perf_counter() - t0


def oracle_chain(y, C, st, seed):
    from gpirt_amd import _abi
    from oracle import oracle
    n, m, H = y.shape
    cfg = _abi.default_config(n=n, m=m, H=H, C=C, sample_iterations=SAMPLE_ITERS,
                              burn_iterations=BURNOUT_ITERS, thin=THIN, seed=seed, theta_os=st["theta_os"],
                              theta_ls=st["theta_ls"], flags=_abi.RETURN_FSTAR)
    t0 = time.perf_counter()
    r = oracle.run(cfg, y, theta_init=st["theta_init"], beta_prior_means=st["bpm"], beta_prior_sds=st["bps"])
    ch = dict(theta=np.moveaxis(r["theta"], -1, 0), fstar=np.moveaxis(r["fstar"], -1, 0),
              beta=np.moveaxis(r["beta"], -1, 0), threshold=np.moveaxis(r["threshold"], -1, 0))
    return ch, time.perf_counter() - t0


def metrics(ch, y, train_idx, theta, truth):
    from gpirt_amd import post
    sm = post.summarize_draws(ch, drop_first_theta=True)       # samples$theta[-1, i, h], :123
    ev = post.evaluate(sm, y, train_idx, theta, truth)
    return {k: float(ev[k]) for k in METRICS}, sm["theta_mean"]


def run_fixture(stem, C):
    y, train_idx, theta, truth = load(stem)
    st = settings(y, theta)
    dev, t_dev = device_chain(y, C, st, 1)
    cpu, t_cpu = oracle_chain(y, C, st, 1)
    m_dev, th_dev = metrics(dev, y, train_idx, theta, truth)
    m_cpu, th_cpu = metrics(cpu, y, train_idx, theta, truth)
    # how long the two chains stay on the same path (kept draws are THIN sweeps apart)
    same = np.abs(np.asarray(dev["theta"]) - cpu["theta"]).reshape(len(cpu["theta"]), -1).max(axis=1) < 1e-6
    first_split = int(np.argmin(same)) if not same.all() else len(same)
    seeds = [metrics(device_chain(y, C, st, 1 + k)[0], y, train_idx, theta, truth)[0] for k in range(1, N_SEEDS + 1)]
    sd = {k: float(np.std([m_dev[k]] + [s[k] for s in seeds], ddof=1)) for k in METRICS}
    tol = {k: max(K_SIGMA * sd[k], FLOORS[k]) for k in METRICS}
    return dict(fixture=stem, C=C, n=int(y.shape[0]), m=int(y.shape[1]), H=int(y.shape[2]),
                sweeps=dict(sample=SAMPLE_ITERS, burn=BURNOUT_ITERS, thin=THIN),
                device=m_dev, cpu_oracle=m_cpu, sd_over_device_seeds=sd, tolerance=tol,
                abs_diff={k: abs(m_dev[k] - m_cpu[k]) for k in METRICS},
                cor_theta_hat_cpu_vs_device=float(np.corrcoef(th_dev.ravel(), th_cpu.ravel())[0, 1]),
                kept_draws_identical_to_1e6=first_split, kept_draws=int(len(same)),
                seconds=dict(device=round(t_dev, 2), cpu_oracle=round(t_cpu, 2)),
                device_diagnostics={k: (float(v) if isinstance(v, float) else int(v))
                                    for k, v in dev["diagnostics"].items()})


def run_mid_rung(n=2500, m=1024, C=5, burn=100, sample=100, seed=1):
    """A rung between the reference's n = 100 data files and the scale-out size, where the device takes
    its large-size code paths (a workgroup per unit in the slice kernels, the tabulated and bucketed
    respondent step, several super-panels in the factorisation): synthetic data by the recipe of
    generate_data.R:36-48,92-119, `theta_init = theta + 0.05 N(0,1)` (gpirt-sythetic.R:86), burn + sample
    sweeps on the device and on the CPU oracle with the same streams, compared sweep by sweep."""
    from gpirt_amd import _abi
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate
    from oracle import oracle
    d = generate(n, m, C, seed=seed)
    init = d["theta"] + 0.05 * np.random.default_rng(seed).normal(size=d["theta"].shape)
    cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=seed)
    g, c = Sampler(cfg, d["y"], theta_init=init), oracle.Sampler(cfg, d["y"], theta_init=init)
    sum_g, sum_c = np.zeros(n), np.zeros(n)
    first_split, worst = None, 0.0
    t_dev = t_cpu = 0.0
    for it in range(burn + sample):
        t0 = time.perf_counter(); g.sweep(1); t_dev += time.perf_counter() - t0
        t0 = time.perf_counter(); c.sweep(1); t_cpu += time.perf_counter() - t0
        tg, tc = g.state(f=False, fstar=False)["theta"][:, 0], c.state()["theta"][:, 0]
        diff = float(np.abs(tg - tc).max())
        if first_split is None:
            worst = max(worst, diff)
            if diff > 1e-6:
                first_split = it + 1
        if it >= burn:
            sum_g += tg
            sum_c += tc
        if (it + 1) % 20 == 0:
            print("mid rung: sweep %d of %d, cpu %.1f s" % (it + 1, burn + sample, t_cpu), file=sys.stderr, flush=True)
    dg, dc = g.diagnostics(), c.diagnostics()
    g.close(); c.close()
    th_g, th_c = sum_g / sample, sum_c / sample
    return dict(rung="synthetic n=%d m=%d C=%d H=1" % (n, m, C), sweeps=dict(burn=burn, sample=sample),
                cor_theta_hat_cpu_vs_device=float(np.corrcoef(th_g, th_c)[0, 1]),
                cor_theta_hat_device_vs_generating=float(np.corrcoef(th_g, d["theta"][:, 0])[0, 1]),
                cor_theta_hat_cpu_vs_generating=float(np.corrcoef(th_c, d["theta"][:, 0])[0, 1]),
                max_abs_diff_theta_hat=float(np.abs(th_g - th_c).max()),
                first_sweep_with_theta_apart_by_1e6=first_split,
                max_abs_theta_diff_while_together=worst,
                device_diagnostics={k: int(v) for k, v in dg.items()},
                cpu_diagnostics={k: int(v) for k, v in dc.items()},
                diagnostics_equal=dg == dc,
                seconds=dict(device=round(t_dev, 2), cpu_oracle=round(t_cpu, 2)))


def check_mid_rung(row):
    bad = []
    if not row["cor_theta_hat_cpu_vs_device"] >= 0.99:
        bad.append("%s: cor(theta_hat cpu, device) = %.4f < 0.99" % (row["rung"], row["cor_theta_hat_cpu_vs_device"]))
    if row["first_sweep_with_theta_apart_by_1e6"] is None and not row["diagnostics_equal"]:
        bad.append("%s: the chains stayed together but their evaluation counters differ" % row["rung"])
    if row["device_diagnostics"]["ess_capped"] != 0:
        bad.append("%s: capped slice updates on the device" % row["rung"])
    return bad


def check(row):
    """The assertions of the parity bar; returns the list of violations."""
    bad = []
    for k in METRICS:
        if not row["abs_diff"][k] <= row["tolerance"][k]:
            bad.append("%s %s: cpu %.4f device %.4f (tolerance %.4f)" %
                       (row["fixture"], k, row["cpu_oracle"][k], row["device"][k], row["tolerance"][k]))
    if not row["cor_theta_hat_cpu_vs_device"] >= 0.99:
        bad.append("%s cor(theta_hat cpu, device) = %.4f < 0.99" % (row["fixture"], row["cor_theta_hat_cpu_vs_device"]))
    if row["device_diagnostics"]["ess_capped"] != 0:
        bad.append("%s: %d capped slice updates" % (row["fixture"], row["device_diagnostics"]["ess_capped"]))
    return bad


if __name__ == "__main__":
    rows = []
    for stem, C in FIXTURES:
        rows.append(run_fixture(stem, C))
        print("fixture %s done: %s" % (stem, rows[-1]["seconds"]), file=sys.stderr, flush=True)
    mid = run_mid_rung()
    table = dict(settings="gpirt-sythetic.R:60-94: 500 + 500 sweeps, THIN 4, theta_init = theta + 0.05 N(0,1), "
                          "theta_ls = H/2, RBF, thresholds sampled, constant_IRF = 0",
                 metrics=list(METRICS), tolerance_rule="max(%g x sd over %d device seeds, floor)" % (K_SIGMA, N_SEEDS + 1),
                 rows=rows, mid_rung=mid, violations=[b for r in rows for b in check(r)] + check_mid_rung(mid))
    text = json.dumps(table, indent=1)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write(text + "\n")
    for r in rows:
        print(r["fixture"], "cor(cpu,device) %.4f" % r["cor_theta_hat_cpu_vs_device"],
              "identical kept draws %d/%d" % (r["kept_draws_identical_to_1e6"], r["kept_draws"]), r["seconds"])
        for k in METRICS:
            print("   %-20s cpu %9.4f  device %9.4f  |diff| %.4f  tol %.4f" %
                  (k, r["cpu_oracle"][k], r["device"][k], r["abs_diff"][k], r["tolerance"][k]))
    print(mid["rung"], "cor(cpu,device) %.5f" % mid["cor_theta_hat_cpu_vs_device"], "first split at sweep",
          mid["first_sweep_with_theta_apart_by_1e6"], "counters equal", mid["diagnostics_equal"], mid["seconds"])
    print("violations:", table["violations"])
    sys.exit(1 if table["violations"] else 0)
