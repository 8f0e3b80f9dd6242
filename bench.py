#!/usr/bin/env python3
"""Gibbs sweeps/sec of the GPIRT sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one Gibbs sweep (main.m:46-87 body) over the whole synthetic response matrix of
the scale-out configuration: n = 10 000 respondents, m = 4096 items, C = 5
ordinal categories, H = 1.  With N > 1 the items are sharded over the ranks (one process per
GPU, `python -m torch.distributed.run ...`), the total work is fixed (strong scaling) and the
respondent step exchanges one n-vector all-reduce per proposal round over RCCL.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (vendor data sheet; 256 CUs x 128 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
SETUP_SWEEPS = 13


def dense_flops(n, m, G):
    """Algorithmic FP64 flop per sweep of the dense part: potrf n^3/3, nu = L Z n^2 m,
    f* solves 2 n^2 m, K* alpha 2 G n m."""
    return dict(potrf=n ** 3 / 3.0, nu=float(n) * n * m, fstar=2.0 * n * n * m + 2.0 * G * n * m)


def substeps(n, m, C, G, ms, evals, r0=0, lookahead=False, k0=0):
    """Achieved rate of every sub-step of the sweep from its ALGORITHMIC work (the counts of the
    reference's formulation: Cholesky n^3/3, L*Z n^2 m, f* by two triangular solves per item
    2 n^2 m + 2 G n m, a likelihood pass n m cells of 8 B f + 1 B y, an f proposal 17 B/cell)
    and its measured time.  The f* step executes far fewer flop than that count (a kernel
    reproduced from 128 nodes turns it into F^T Phi, a Gram matrix and thin products), so its
    rate can exceed the matrix peak; `executed_flop` says what was actually run."""
    def rate(flop, t):
        return flop / (t * 1e-3) / 1e12 if t > 0 else 0.0
    cells_f, cells_p, cells_t = evals["ess_f_rounds"] * n, evals["ess_param_rounds"] * n, evals["ess_theta_rounds"] * n * m
    fstar_exec = 4.0 * n * 128 * 128 + 2.0 * n * m * 128 + 2.0 * 128 * G * m
    # r0 rows of nu = L*Z are computed on a second stream beside the factorisation: their flop
    # (r0^2 m) fall into the "potrf" span, the launch after it covers rows r0..n
    # (and the first k0 columns of L's share of the rows r0..n as well: 2 k0 (n - r0) m)
    fl_tail = 2.0 * k0 * (n - r0) * m
    fl_beside, fl_after = float(r0) * r0 * m + fl_tail, (float(n) * n - float(r0) * r0) * m - fl_tail
    out = {
        "potrf": {"ms": ms["potrf"], "algorithmic_flop": n ** 3 / 3.0 + fl_beside,
                  "tflops": rate(n ** 3 / 3.0 + fl_beside, ms["potrf"]),
                  "rows_of_nu_computed_beside_it": r0, "columns_of_L_added_to_the_other_rows_beside_it": k0,
                  "lookahead": lookahead},
        "nu=L*Z": {"ms": ms["nu"], "algorithmic_flop": fl_after, "tflops": rate(fl_after, ms["nu"]),
                   "rows": [r0, n], "from_column_of_L": k0},
        "fstar": {"ms": ms["fstar"], "algorithmic_flop": 2.0 * n * n * m + 2.0 * G * n * m,
                  "executed_flop": fstar_exec, "tflops_algorithmic": rate(2.0 * n * n * m + 2.0 * G * n * m, ms["fstar"]),
                  "tflops_executed": rate(fstar_exec, ms["fstar"])},
        "k_build": {"ms": ms["k_build"], "algorithmic_bytes": 8.0 * n * n,
                    "GBps": 8.0 * n * n / (ms["k_build"] * 1e-3) / 1e9 if ms["k_build"] > 0 else 0.0},
    }
    for name, cells, bpc in (("ess_f", cells_f, 17.0), ("ess_param", cells_p, 9.0), ("theta", cells_t, 9.0)):
        t = ms[name]
        out[name] = {"ms": t, "cells": cells, "Gcells_per_s": cells / (t * 1e-3) / 1e9 if t > 0 else 0.0,
                     "algorithmic_GBps": cells * bpc / (t * 1e-3) / 1e9 if t > 0 else 0.0}
    # the respondent step counts every respondent in every lock-step round, although those
    # that have accepted drop out: its cell count is an upper bound
    out["theta"]["cells_are_upper_bound"] = True
    return out


PMC_FILE = "r03_pmc_kernels.json"


def pmc_record():
    """The committed rocprofv3 PMC passes (profiles/PMC_FILE: FETCH_SIZE doubled as the
    gfx950 correction asks, plus WRITE_SIZE, separate passes; counters cannot be read from inside
    this process).  Valid for the benchmark's own configuration on one GPU only."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
    except OSError:
        return None


def pmc_kernel(rec, label, n, m, world):
    if rec is None or world != 1 or (rec.get("config", {}).get("n"), rec.get("config", {}).get("m")) != (n, m):
        return None
    return rec["kernels"].get(label)


def cpu_baseline(args, n, m, G):
    """The same sweep on the host's cores, at the benchmark's size.

    Dense phases (Cholesky of the n x n kernel matrix, nu = L Z for all items, the two
    triangular solves and the grid product of the f* step) are timed at FULL size with the
    LAPACK / BLAS that numpy and scipy carry (multithreaded OpenBLAS): what the reference's
    Armadillo code bottoms out in, given the best case of one factorisation per sweep and all
    items batched.  The likelihood phases (slice updates of f, of beta / thresholds and of
    theta, the snap) are timed with the CPU oracle (OpenMP over items / respondents) at the
    full n and a share `--cpu-m` of the items, and scaled by m / cpu_m: they are linear in the
    number of items.  Nothing is extrapolated in n."""
    import scipy.linalg as sl
    from gpirt_amd import _abi
    from gpirt_amd.synth import generate
    from oracle import oracle
    try:
        from threadpoolctl import threadpool_info
        blas = [(i.get("internal_api"), i.get("num_threads")) for i in threadpool_info() if i.get("user_api") == "blas"]
    except Exception:                                    # noqa: BLE001
        blas = []
    rng = np.random.default_rng(1)
    x = rng.normal(size=n)
    t = {}
    t0 = time.perf_counter()
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2)
    K[np.diag_indices(n)] += 1e-6
    t["k_build"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    L = sl.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    t["potrf"] = time.perf_counter() - t0
    del K
    Z = rng.normal(size=(n, m))
    t0 = time.perf_counter()
    F = sl.blas.dtrmm(1.0, L, Z, lower=1)                # nu = L Z, triangular count n^2 m
    t_trmm = time.perf_counter() - t0
    t0 = time.perf_counter()
    F = L @ Z                                            # the same product as a full dgemm (twice the
    t_gemm = time.perf_counter() - t0                    # flop, but the better-threaded routine)
    t["nu"] = min(t_trmm, t_gemm)
    del Z
    xs = -5.0 + 0.01 * np.arange(G)
    t0 = time.perf_counter()
    alpha = sl.cho_solve((L, True), F, overwrite_b=True, check_finite=False)
    Ks = np.exp(-0.5 * (xs[:, None] - x[None, :]) ** 2)
    fstar = Ks @ alpha
    t["fstar"] = time.perf_counter() - t0
    fstar_ref = fstar
    # The same f* by the formulation the GPU runs (the kernel reproduced from 128 Chebyshev nodes in both
    # arguments, then the push-through identity: no solve with the n x n factor), on the same BLAS: the
    # like-for-like CPU time of that phase.  Set-up that depends on the nodes only (eigen-decomposition of
    # K(c, c), B^T Lag) is left out of the time, as on the GPU, where it is done once at creation.
    R = 128
    lo, hi = float(xs[0]), float(xs[-1])
    c = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * np.arange(R) / (R - 1))
    w = np.where(np.arange(R) % 2 == 1, -1.0, 1.0)
    w[0] *= 0.5
    w[-1] *= 0.5

    def basis(pts):
        d = pts[:, None] - c[None, :]
        hit = d == 0.0
        d[hit] = 1.0
        b = w[None, :] / d
        b /= b.sum(axis=1, keepdims=True)
        rows = hit.any(axis=1)
        b[rows] = hit[rows].astype(float)
        return b
    ev, V = np.linalg.eigh(np.exp(-0.5 * (c[:, None] - c[None, :]) ** 2))
    keep = ev > 1e-17 * ev.max()
    B = V[:, keep] * np.sqrt(ev[keep])
    BL = B.T @ basis(xs).T                                # [rank, G]
    t0 = time.perf_counter()
    Phi = basis(x) @ B                                    # [n, rank]
    M = Phi.T @ Phi
    M[np.diag_indices(M.shape[0])] += 1e-6
    fstar_lr = (sl.cho_solve(sl.cho_factor(M, lower=True), (F.T @ Phi).T, check_finite=False).T @ BL).T
    t["fstar_lowrank"] = time.perf_counter() - t0
    fstar_gap = float(np.abs(fstar_lr - fstar_ref).max() / max(1.0, np.abs(fstar_ref).max()))
    del L, F, alpha, Ks, fstar, fstar_ref, fstar_lr, Phi
    ms = min(args.cpu_m, m)
    d = generate(n, ms, 5, seed=1)
    cfg = _abi.default_config(n=n, m=ms, H=1, C=5, seed=1)
    s = oracle.Sampler(cfg, d["y"], theta_init=d["theta"])
    s.sweep(args.cpu_sweeps)
    ph = {k: v / args.cpu_sweeps for k, v in s.phase_seconds().items()}
    s.close()
    lik = {k: ph[k] * (m / ms) for k in ("ess_f", "param", "theta", "snap")}
    total = t["k_build"] + t["potrf"] + t["nu"] + t["fstar"] + sum(lik.values())
    total_lr = total - t["fstar"] + t["fstar_lowrank"]
    fl = dense_flops(n, m, G)
    return {
        "value": 1.0 / total, "unit": "sweeps/s", "cores": oracle.num_threads(), "kind": "port",
        # the same sweep with the f* step in the formulation the GPU runs (phases_s.fstar_lowrank in place of
        # phases_s.fstar): what a like-for-like ratio is quoted against
        "value_with_lowrank_fstar": 1.0 / total_lr,
        "fstar_lowrank_vs_solve_max_rel_diff": fstar_gap,
        "sample": ("CPU restatement, not the Rcpp package: ONE sweep's work at the full size n=%d, m=%d, C=5. "
                   "Dense phases at full size on LAPACK/BLAS (%s): k_build %.2f s, dpotrf %.2f s (%.0f GFLOP/s), "
                   "nu = L Z %.2f s (faster of dtrmm and dgemm, %.0f GFLOP/s of its n^2 m), f* = two triangular solves + grid product %.2f s "
                   "(the reference's formulation, counted in `value`; the low-rank identity the GPU runs takes %.2f s on the same BLAS: "
                   "`value_with_lowrank_fstar`). "
                   "Likelihood phases by oracle/gpirt_oracle.c (OpenMP, %d threads) over %d sweep(s) at n=%d with "
                   "%d of the %d items, scaled by %g (linear in items): slice f %.2f s, beta/thresholds %.2f s, "
                   "theta %.2f s, snap %.2f s."
                   % (n, m, blas or "threads unknown", t["k_build"], t["potrf"], fl["potrf"] / t["potrf"] / 1e9,
                      t["nu"], fl["nu"] / t["nu"] / 1e9, t["fstar"], t["fstar_lowrank"], oracle.num_threads(), args.cpu_sweeps, n, ms, m,
                      m / ms, lik["ess_f"], lik["param"], lik["theta"], lik["snap"])),
        "seconds_per_sweep": total,
        "phases_s": {**{k: round(v, 3) for k, v in t.items()}, **{k: round(v, 3) for k, v in lik.items()}},
        "blas_threads": blas,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--respondents", dest="n", type=int, default=10000)
    ap.add_argument("--m", "--items", dest="m", type=int, default=4096)
    ap.add_argument("--C", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--cpu-m", type=int, default=256, help="items in the oracle's timed sample of the likelihood phases")
    ap.add_argument("--cpu-sweeps", type=int, default=1)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # the exchange of the sharded respondent step: by default the library's own RCCL collective
    # (libgpirt_rccl.so, no torch on the data path; torch.distributed's gloo group only carries
    # the communicator's 128-byte identifier, the barriers and the MAX of the ranks' clocks).
    # GPIRT_BENCH_BACKEND=gloo rehearses the multi-rank logic on a box with fewer GPUs than
    # ranks (exchange staged through host memory); GPIRT_BENCH_COLLECTIVE=torch selects the
    # torch.distributed "nccl" hook instead.
    rehearsal = os.environ.get("GPIRT_BENCH_BACKEND") == "gloo"
    host_staged = os.environ.get("GPIRT_BENCH_BACKEND") == "host"   # gloo + the library's own HIP runtime, no RCCL
    use_torch_nccl = world > 1 and not rehearsal and os.environ.get("GPIRT_BENCH_COLLECTIVE") == "torch"
    if world > 1 and (rehearsal or use_torch_nccl):
        # torch first: it brings its own copy of the HIP runtime, and a process in which
        # libgpirt_hip.so has already initialised the system's copy finds no GPU from torch
        import torch
        if rehearsal:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        torch.zeros(1, device="cuda")

    from gpirt_amd import _abi, _lib
    from gpirt_amd.dist import host_staged_allreduce, item_range, torch_allreduce
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate

    lib = _lib.lib()
    if lib.gpirt_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: " + "no HIP device visible")
    if host_staged:
        local_rank = local_rank % lib.gpirt_device_count()
    _lib.check(lib.gpirt_set_device(local_rank))

    dist = None
    hook = None
    native = None
    fallback_reason = None
    on_stream = False
    torch_on_gpu = False
    if world > 1:
        import torch
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
            hook = torch_allreduce(device_buffers=True, via_host=True)
            torch_on_gpu = True
        elif use_torch_nccl:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            on_stream = not os.environ.get("GPIRT_BENCH_SYNC_ALLREDUCE")
            hook = torch_allreduce(device_buffers=True, on_stream=on_stream)
            torch_on_gpu = True
        else:
            dist.init_process_group(backend="gloo")
            why = "GPIRT_BENCH_BACKEND=host"
            if not host_staged:
                # the communicator is created by all ranks or by none (the ranks vote over gloo before
                # RCCL's own bootstrap, which is a collective): a rank whose RCCL is missing or whose
                # bootstrap fails takes every rank to the host-staged hook (a slower line, not none)
                from gpirt_amd import comm as gcomm
                native, why = gcomm.create_all_or_none(rank, world)
            if native is None:
                fallback_reason = why
                hook = host_staged_allreduce()
                if rank == 0 and not host_staged:
                    print("bench.py: no RCCL communicator (%s); the exchange goes through host memory" % why,
                          file=sys.stderr)

    n, m, C = args.n, args.m, args.C
    data = generate(n, m, C, seed=args.seed)          # identical on every rank
    rng = np.random.default_rng(args.seed)
    theta_init = data["theta"] + 0.05 * rng.normal(size=data["theta"].shape)   # gpirt-sythetic.R:86
    cfg = _abi.default_config(n=n, m=m, H=1, C=C, seed=args.seed, theta_os=1.0, theta_ls=1.0,
                              rank=rank, world=world)
    cfg.flags |= int(os.environ.get("GPIRT_BENCH_FLAGS", "0"))   # e.g. 8 | 32: variants for profiling
    keep = None
    if world > 1:
        cfg.item_begin, cfg.item_end = item_range(m, rank, world)
        if native is not None:
            native.bind(cfg)              # rank, world, the RCCL hook, stream-ordered
        else:
            keep = _abi.ALLREDUCE_FN(lambda ptr, count, user: hook(ptr, count))
            cfg.allreduce = keep
            if on_stream:
                cfg.flags |= _abi.ALLREDUCE_ON_STREAM
    s = Sampler(cfg, data["y"], theta_init=theta_init, beta_prior_means=np.zeros((3, m)),
                beta_prior_sds=np.ones((3, m)))
    if hook is not None:
        hook.stream_ptr = s.stream()
    if native is not None:
        native.set_stream(s.stream())

    def barrier():
        # sweep() drains the sampler's streams before it returns: nothing is in flight here
        if dist is not None:
            if torch_on_gpu:
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    def max_over_ranks(v):
        if dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # the sampler times its first sweeps of the dense route (three per schedule, up to four
    # schedules) to choose whether the first rows of nu = L*Z run beside the factorisation and
    # whether the factorisation runs its trailing updates beside the next block columns' steps
    # (same chain bit for bit): that choice is part of the set-up, before the warm-up steps.  Its
    # second sweep runs the plain schedule, the whole product as one launch after the
    # factorisation: those times are kept for the roofline report.
    s.sweep(1)
    pa = s.phase_times()
    s.sweep(1)
    pb = s.phase_times()
    alone = s.nu_rows_beside_factor() == 0       # sweep 2 of the set-up runs nothing beside the factorisation
    nu_alone_ms = pb["nu"][0] - pa["nu"][0] if alone else None
    potrf_alone_ms = pb["potrf"][0] - pa["potrf"][0] if alone else None
    # the eighth set-up sweep (where the sampler has that schedule to offer) is the factorisation with
    # look-ahead and nothing else beside it
    s.sweep(5)
    pc = s.phase_times()
    s.sweep(1)
    pd = s.phase_times()
    potrf_ahead_ms = pd["potrf"][0] - pc["potrf"][0] if s.factor_lookahead() and s.nu_rows_beside_factor() == 0 else None
    s.sweep(SETUP_SWEEPS - 8)
    s.sweep(args.warmup)
    ph0 = s.phase_times()
    d0 = s.diagnostics()
    up0 = s.update_times()
    calls0 = native.calls() if native is not None else 0
    barrier()
    t0 = time.perf_counter()
    s.sweep(args.steps)                  # drains its stream before returning
    barrier()
    elapsed = time.perf_counter() - t0
    calls1 = native.calls() if native is not None else 0
    ph1 = s.phase_times()
    d1 = s.diagnostics()
    up1 = s.update_times()
    r0 = min(s.nu_rows_beside_factor(), n)
    lookahead = s.factor_lookahead()
    k0 = min(512 * s.nu_tail_panels(), r0)
    elapsed = max_over_ranks(elapsed)
    # With the slice steps in two shares of the units (the second on a stream of its own, the f* step beside its
    # parameter update) the marks of "ess_f", "ess_param" and "fstar" on the main stream no longer bracket one
    # step each: the same sweeps again with one launch per step give each step's own time, for the rates below.
    shares = s.slice_second_share()
    alone = None
    if shares > 0:
        kept = os.environ.get("GPIRT_SLICE_SHARES")
        os.environ["GPIRT_SLICE_SHARES"] = "0"
        a0, e0 = s.phase_times(), s.diagnostics()
        s.sweep(args.steps)
        a1, e1 = s.phase_times(), s.diagnostics()
        alone_evals = {k: (e1[k] - e0[k]) / args.steps for k in ("ess_f_rounds", "ess_param_rounds")}
        alone = {k: (a1[k][0] - a0[k][0]) / args.steps for k in a1}
        if kept is None:
            del os.environ["GPIRT_SLICE_SHARES"]
        else:
            os.environ["GPIRT_SLICE_SHARES"] = kept

    # The same sweep with the prior factor kept in generator form (GPIRT_GENERATOR_PRIOR): the
    # headline above is the dense formulation (n x n Cholesky, nu = L*Z as one triangular
    # product); this is the same chain to ~1e-8 without the n x n matrix, reported beside it.
    variant = None
    if not args.no_variants and not (cfg.flags & _abi.GENERATOR_PRIOR):
        s.close()
        cfg.flags |= _abi.GENERATOR_PRIOR
        s = Sampler(cfg, data["y"], theta_init=theta_init, beta_prior_means=np.zeros((3, m)),
                    beta_prior_sds=np.ones((3, m)))
        if hook is not None:
            hook._ext = None
            hook.stream_ptr = s.stream()
        if native is not None:
            native.set_stream(s.stream())
        s.sweep(args.warmup)
        v0 = s.phase_times()
        barrier()
        tv = time.perf_counter()
        s.sweep(args.steps)
        barrier()
        v_elapsed = time.perf_counter() - tv
        v1 = s.phase_times()
        v_elapsed = max_over_ranks(v_elapsed)
        variant = {"flags": "GPIRT_GENERATOR_PRIOR", "value": args.steps / v_elapsed, "unit": "sweeps/s",
                   "ms_per_step": 1e3 * v_elapsed / args.steps,
                   "phases_ms": {k: (v1[k][0] - v0[k][0]) / args.steps for k in v1},
                   "note": "same sweep, same inputs; the Cholesky factor of K + s2 I is kept as 128 x 128 "
                           "diagonal blocks plus generators (block semiseparable) instead of an n x n "
                           "matrix; held to the same parity tests as the dense route"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        phases = {k: (ph1[k][0] - ph0[k][0]) / args.steps for k in ph1}
        launches = {k: (ph1[k][1] - ph0[k][1]) / args.steps for k in ph1}
        mloc = (cfg.item_end - cfg.item_begin) if world > 1 else m
        fl = dense_flops(n, mloc, cfg.n_grid)
        # Dominant kernel by device time: the FP64 MFMA GEMM.  Two of its instances carry the dense
        # work of a sweep: the rank-512 trailing updates of the Cholesky (gemm_direct_kernel<true,1>,
        # ~19 launches per sweep, each bracketed by HIP events on the sampler's stream during the
        # timed region) and nu = L*Z (phase "nu": the one launch per sweep after the factorisation,
        # timed the same way; the rows computed beside the factorisation run as chunks on a second
        # stream and are inside the "potrf" span).
        rec = pmc_record()
        nu_ms = phases["nu"]
        tiles = -(-n // 128) * -(-mloc // 128)
        nu_kernel = ("gemm_direct_kernel<false,5> (nu = L*Z, rows %d..%d, columns of L from %d on: one launch per "
                     "sweep after the factorisation; rows 0..%d and the earlier columns' share of the other rows run "
                     "beside the factorisation on a second stream as gemm_direct_lone_kernel<false,2> / <false,0>)"
                     % (r0, n, k0, r0) if r0 > 0 else
                     "gemm_direct_kernel<false,4> (nu = L*Z, one launch per sweep)" if tiles >= 2048 else
                     "gemm_direct_kernel<false,2> (nu = L*Z, one launch per sweep%s)"
                     % (", K in %d slabs + sum_slabs_kernel" % (8 if tiles < 512 else 4) if tiles < 1024 else ""))
        # triangular count of the rows r0..n, less the first k0 columns of L, whose share those rows
        # took beside the factorisation
        fl_launch = (float(n) * n - float(r0) * r0 - 2.0 * k0 * (n - r0)) * mloc
        nu_achieved = fl_launch / (nu_ms * 1e-3) / 1e12 if nu_ms > 0 else 0.0
        upd_ms, upd_n, upd_fl = up1[0] - up0[0], up1[1] - up0[1], up1[2] - up0[2]
        upd_achieved = upd_fl / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        upd_pmc = None if lookahead else pmc_kernel(rec, "syrk", n, m, world)
        nu_pmc = pmc_kernel(rec, "nu_rest" if r0 > 0 else "nu_whole", n, m, world)
        evals = {k: (d1[k] - d0[k]) / args.steps for k in ("ess_f_rounds", "ess_param_rounds", "ess_theta_rounds")}
        dense_ms = phases["potrf"] + phases["nu"]     # the f* step no longer runs its algorithmic count
        out = {
            "metric": "Gibbs sweeps/sec at n=10k,m=4096,C=5",
            "value": args.steps / elapsed,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "setup_sweeps": SETUP_SWEEPS,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "synthetic ordinal GPIRT, n=%d respondents, m=%d items, C=%d, H=1, "
                                   "1001-point grid, thresholds and beta sampled" % (n, m, C),
                       "items_per_gpu": mloc, "parallelism": "items sharded x%d" % world,
                       "collective": ("none" if world == 1 else "libgpirt_rccl.so: ncclAllGather + sum in rank order, "
                                      "stream-ordered" if native is not None else
                                      "gloo through host memory (rehearsal)" if rehearsal else
                                      "gloo through host memory (%s)" % fallback_reason if fallback_reason else
                                      "torch.distributed nccl"),
                       # what RCCL itself says about the communicator (ncclCommCount), and how often a sweep used it
                       "collective_world": native.world_reported() if native is not None else None,
                       "collective_calls_per_step": (calls1 - calls0) / args.steps if native is not None else None,
                       "collective_doubles_per_call_max": native.capacity() if native is not None else None},
            "roofline": {
                "kernel": ("gemm_direct_lone_kernel<true,1> (rank-512 trailing updates of the blocked Cholesky, lower "
                           "tiles only, on the second stream beside the next block columns' steps: fixed grid, one "
                           "workgroup to a CU)" if lookahead else
                           "gemm_direct_kernel<true,1> (rank-512 trailing updates of the blocked Cholesky, lower "
                           "tiles only): the kernel with the largest share of device time"),
                "bound": "mfma", "achieved": upd_achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": upd_achieved / FP64_MFMA_PEAK_TFLOPS,
                "launches_per_step": upd_n / args.steps if args.steps else 0,
                "flop_per_launch": upd_fl / upd_n if upd_n else 0.0,
                "ms_per_launch": upd_ms / upd_n if upd_n else 0.0,
                "ms_per_step": upd_ms / args.steps if args.steps else 0.0,
                "algorithmic_bytes_per_launch": ((upd_fl / upd_n) / (2.0 * 128 * 128 * 512) * (2 * 128 * 128 * 8)
                                                 if upd_n else 0.0),
                "traffic": upd_pmc.get("hbm_bytes_per_launch") if upd_pmc else None,
                "traffic_note": "mean HBM bytes per launch of this kernel from profiles/" + PMC_FILE +
                                " (rocprofv3 --pmc, 2 x FETCH_SIZE + WRITE_SIZE); algorithmic = read + write of "
                                "each lower tile of the trailing matrix once (the 512-wide panel is re-read from cache)",
                "other_instances": {
                    "nu_after_factorisation": {
                        "kernel": nu_kernel, "achieved": nu_achieved, "frac": nu_achieved / FP64_MFMA_PEAK_TFLOPS,
                        "flop_per_launch": fl_launch, "ms_per_launch": nu_ms,
                        "algorithmic_bytes_per_launch": 8.0 * ((float(n) * n - float(r0) * r0) / 2 - float(k0) * (n - r0)
                                                               + float(n - k0) * mloc + (2.0 if k0 else 1.0) * (n - r0) * mloc),
                        "traffic": nu_pmc.get("hbm_bytes_per_launch") if nu_pmc else None,
                        "note": ("this launch shares the chip with the chunk of rows that the second stream has not "
                                 "finished when the factorisation ends (224 workgroups on 224 CUs for about half of "
                                 "the launch's span in the kernel trace): `achieved` counts this launch's flop only, "
                                 "the two together run at ~0.8 of peak over the span (DESIGN.md, section 3)") if r0 > 0 else None},
                },
            },
            "substeps": substeps(n, mloc, C, cfg.n_grid, dict(phases, **{k: alone[k] for k in ("ess_f", "ess_param", "fstar")})
                                 if alone else phases, dict(evals, **alone_evals) if alone else evals, r0, lookahead, k0),
            "phases_ms": phases,
            "launch_marks_per_step": launches,
            "dense_tflops": (fl["potrf"] + fl["nu"]) / (dense_ms * 1e-3) / 1e12 if dense_ms > 0 else 0.0,
            "likelihood_evals_per_step": evals,
            "diagnostics": d1,
        }
        if nu_alone_ms:
            whole = float(n) * n * mloc / (nu_alone_ms * 1e-3) / 1e12
            wp = pmc_kernel(rec, "nu_whole", n, m, world)
            out["roofline"]["other_instances"]["nu_whole_product_in_one_launch"] = {
                "kernel": "gemm_direct_kernel<false,4>, sweep 2 of the set-up (nothing beside the factorisation)",
                "ms_per_launch": nu_alone_ms, "flop_per_launch": float(n) * n * mloc,
                "achieved": whole, "frac": whole / FP64_MFMA_PEAK_TFLOPS,
                "algorithmic_bytes_per_launch": 8.0 * (float(n) * n / 2 + 2.0 * n * mloc),
                "traffic": wp.get("hbm_bytes_per_launch") if wp else None}
        if potrf_alone_ms:
            ch = n ** 3 / 3.0 / (potrf_alone_ms * 1e-3) / 1e12
            out["roofline"]["cholesky_phase_alone"] = {
                "what": "the whole factorisation with nothing beside it (sweep 2 of the set-up): diagonal-block "
                        "kernels, panel solves, in-panel updates and trailing updates, n^3/3 flop over its span",
                "ms": potrf_alone_ms, "achieved": ch, "frac": ch / FP64_MFMA_PEAK_TFLOPS}
        if potrf_ahead_ms:
            ch = n ** 3 / 3.0 / (potrf_ahead_ms * 1e-3) / 1e12
            out["roofline"]["cholesky_phase_alone_with_lookahead"] = {
                "what": "the same with the bulk of each trailing update on a second stream beside the next block "
                        "columns' steps (sweep 8 of the set-up; the schedule the sampler keeps is named in "
                        "substeps.potrf)",
                "ms": potrf_ahead_ms, "achieved": ch, "frac": ch / FP64_MFMA_PEAK_TFLOPS}
        lik = {}
        for label in ("ess_f", "ess_param", "theta_eval_bucket", "theta_table"):
            k = pmc_kernel(rec, label, n, m, world)
            if k and "hbm_bytes_per_launch" in k and "hbm_GBps_profiled" in k:
                lik[label] = {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                              "hbm_GBps_under_the_profiler": k["hbm_GBps_profiled"],
                              "frac_of_hbm_peak": k["hbm_GBps_profiled"] / HBM_PEAK_GBS,
                              "valu_insts_per_launch": k.get("SQ_INSTS_VALU"),
                              "valu_issue_cycles_share_of_wave_cycles": k.get("valu_active_share_of_wave_cycles")}
        if lik:
            out["likelihood_kernels_measured"] = {
                "source": "profiles/" + PMC_FILE + " (rocprofv3 --pmc passes of this benchmark; durations and "
                          "rates are those under the profiler, which serialises kernels and lowers the clock)",
                "bound": "FP64 VALU issue and LDS gathers of the Mills-ratio table, not HBM: see valu_issue_cycles_share", "kernels": lik}
        if alone:
            out["slice_schedule"] = {
                "second_share_units": shares,
                "phases_ms_one_launch_per_step": {k: alone[k] for k in ("ess_f", "ess_param", "fstar")},
                "span_ms": phases["ess_f"] + phases["ess_param"] + phases["fstar"],
                "likelihood_evals_per_step_in_those_sweeps": alone_evals,
                "note": "the slice updates of f and of (beta, thresholds) run in two shares of the units, the second "
                        "on a low-priority stream, and the f* step runs beside the second share's parameter update: "
                        "in `phases_ms` the marks of ess_f / ess_param / fstar overlap (their sum is the span); "
                        "`substeps` quotes these three from %d more sweeps with one launch per step "
                        "(GPIRT_SLICE_SHARES=0), same chain bit for bit" % args.steps}
        if variant is not None:
            out["variants"] = {"generator_prior": variant}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, n, m, cfg.n_grid)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["gpu_over_cpu_like_for_like"] = out["value"] / out["cpu_baseline"]["value_with_lowrank_fstar"]
        print(json.dumps(out))
    s.close()
    if native is not None:
        native.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
