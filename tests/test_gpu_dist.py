"""The item-sharded DEVICE path: two ranks share the one GPU of the test box (a gloo group
carries the exchange through host memory, because RCCL refuses two ranks on one device), and a
one-rank RCCL group checks the hook the 8-GPU benchmark uses."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, grid, torch_free=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if torch_free:
        # the library first: torch then finds no GPU in this process, and needs none
        from gpirt_amd import _lib
        _lib.check(_lib.lib().gpirt_set_device(0))
    import torch
    import torch.distributed as dist
    from gpirt_amd import _abi
    from gpirt_amd.dist import host_staged_allreduce, item_range, torch_allreduce
    from gpirt_amd.sampler import Sampler
    from conftest import load_fixture

    if not torch_free:
        torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = load_fixture("Wiener_C_5")
    y, theta = _case(d, grid)
    n, m, H = y.shape
    j0, j1 = item_range(m, rank, world)
    hook = host_staged_allreduce() if torch_free else torch_allreduce(device_buffers=True, via_host=True)
    cb = _abi.ALLREDUCE_FN(lambda ptr, count, user: hook(ptr, count))
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_ls=1.5, item_begin=j0, item_end=j1,
                              rank=rank, world=world,
                              theta_mode=_abi.THETA_GRID if grid else _abi.THETA_ESS)
    cfg.allreduce = cb
    s = Sampler(cfg, y, theta_init=theta)
    s.sweep(3)
    st = s.state()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), j0=j0, j1=j1, **st)
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def _case(d, grid):
    """dynamic case: 3 periods; static (grid-mode) case: the first period only"""
    k = 1 if grid else 3
    return d["y"][:, :, :k], d["theta"][:, :k]


@pytest.mark.parametrize("grid,torch_free", [(0, False), (1, False), (0, True)])
def test_two_ranks_on_one_gpu_match_unsharded_device_chain(tmp_path, grid, torch_free):
    """torch_free: the hook that stages through host memory with the library's own HIP runtime
    (gpirt_amd.dist.host_staged_allreduce, the benchmark's fall-back when RCCL cannot start)."""
    import torch.multiprocessing as mp
    from gpirt_amd import _abi
    from gpirt_amd.sampler import Sampler
    from conftest import load_fixture

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), grid, torch_free), nprocs=2, join=True)
    d = load_fixture("Wiener_C_5")
    y, theta = _case(d, grid)
    n, m, H = y.shape
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_ls=1.5,
                              theta_mode=_abi.THETA_GRID if grid else _abi.THETA_ESS)
    ref = Sampler(cfg, y, theta_init=theta)
    ref.sweep(3)
    full = ref.state()
    ref.close()
    for r in range(2):
        p = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        j0, j1 = int(p["j0"]), int(p["j1"])
        assert np.abs(p["theta"] - full["theta"]).max() < 1e-9
        assert np.abs(p["f"] - full["f"][:, j0:j1]).max() < 1e-6
        assert np.abs(p["fstar"] - full["fstar"][:, j0:j1]).max() < 1e-6
        assert np.abs(p["beta"] - full["beta"][:, j0:j1]).max() < 1e-9


def _taps_worker(rank, world, port, out_dir, kernel):
    """One of two ranks on the TAPS cut with the arguments of TAPS/gpirt_TAPS.R:17-36
    (`constant_IRF = 1`: 2460 stacked inputs per item), three of the six questions each."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from gpirt_amd import _abi, _lib
    _lib.check(_lib.lib().gpirt_set_device(0))
    import torch.distributed as dist
    from gpirt_amd.builders import standardised_row_scores
    from gpirt_amd.dist import host_staged_allreduce, item_range
    from gpirt_amd.sampler import Sampler
    from test_gpu_config_d import taps, taps_arguments

    dist.init_process_group("gloo", rank=rank, world_size=world)
    y = np.ascontiguousarray(taps()[:60])
    n, m, H = y.shape
    bps, tps = taps_arguments(n, m)
    j0, j1 = item_range(m, rank, world)
    hook = host_staged_allreduce()
    cb = _abi.ALLREDUCE_FN(lambda ptr, count, user: hook(ptr, count))
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=1, theta_os=1.0, theta_ls=12.0, kernel=kernel,
                              constant_irf=1, item_begin=j0, item_end=j1, rank=rank, world=world)
    cfg.allreduce = cb
    with Sampler(cfg, y, theta_init=standardised_row_scores(y), beta_prior_sds=bps, theta_prior_sds=tps) as s:
        s.sweep(3)
        st, dg = s.state(), s.diagnostics()
    np.savez(os.path.join(out_dir, "taps_k%d_r%d.npz" % (kernel, rank)), j0=j0, j1=j1,
             theta_rounds=dg["ess_theta_rounds"], **st)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kernel", [0, 1], ids=["RBF", "Matern"])
def test_taps_panel_sharded_over_two_ranks_matches_the_unsharded_device_chain(tmp_path, kernel):
    """Dynamic GPIRT on the TAPS panel with the items sharded (TAPS/gpirt_TAPS.R:30-36,
    TAPS/onerun:21-25: `constant_IRF = 1`, both temporal kernels): two ranks sharing the GPU, three
    questions each, against the unsharded device chain, which test_taps_cut_matches_oracle ties to
    the oracle sweep by sweep.  theta is the same chain to rounding (the ranks' partial sums are
    re-associated), each rank's f / f* / beta / thresholds are its slice of the unsharded ones."""
    import torch.multiprocessing as mp
    from gpirt_amd import _abi
    from gpirt_amd.builders import standardised_row_scores
    from gpirt_amd.sampler import Sampler
    from test_gpu_config_d import taps, taps_arguments

    mp.spawn(_taps_worker, args=(2, _free_port(), str(tmp_path), kernel), nprocs=2, join=True)
    y = np.ascontiguousarray(taps()[:60])
    n, m, H = y.shape
    bps, tps = taps_arguments(n, m)
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=1, theta_os=1.0, theta_ls=12.0, kernel=kernel,
                              constant_irf=1)
    with Sampler(cfg, y, theta_init=standardised_row_scores(y), beta_prior_sds=bps, theta_prior_sds=tps) as ref:
        ref.sweep(3)
        full, dg = ref.state(), ref.diagnostics()
    for r in range(2):
        p = np.load(os.path.join(str(tmp_path), "taps_k%d_r%d.npz" % (kernel, r)))
        j0, j1 = int(p["j0"]), int(p["j1"])
        assert (j0, j1) == (3 * r, 3 * r + 3)
        assert int(p["theta_rounds"]) == dg["ess_theta_rounds"]
        assert np.abs(p["theta"] - full["theta"]).max() < 1e-9
        assert np.abs(p["beta"] - full["beta"][:, j0:j1]).max() < 1e-9
        fin = np.isfinite(full["threshold"][j0:j1])
        assert np.abs(p["threshold"][fin] - full["threshold"][j0:j1][fin]).max() < 1e-9
        assert np.abs(p["f"] - full["f"][:, j0:j1]).max() < 1e-5
        assert np.abs(p["fstar"] - full["fstar"][:, j0:j1]).max() < 1e-5


def _nccl_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from gpirt_amd.dist import torch_allreduce
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    hook = torch_allreduce(device_buffers=True)
    assert hook(x.data_ptr(), x.numel()) == 0        # raw device pointer, as the library passes it
    ok = bool(torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64)))
    open(out, "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_rccl_hook_on_a_raw_device_pointer(tmp_path):
    import torch.multiprocessing as mp
    out = os.path.join(str(tmp_path), "r.txt")
    mp.spawn(_nccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert open(out).read() == "ok"


def _onstream_worker(rank, world, port, out_dir):
    """One RCCL rank that plays rank 0 of 2 (it owns the first half of the items): the chain
    is not a valid posterior sample, but the stream-ordered hook and the blocking hook must
    drive it to the same state."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from gpirt_amd import _abi
    from gpirt_amd.dist import item_range, torch_allreduce
    from gpirt_amd.sampler import Sampler
    from conftest import load_fixture

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    d = load_fixture("Wiener_C_5")
    y, theta = d["y"][:, :, :3], d["theta"][:, :3]
    n, m, H = y.shape
    j0, j1 = item_range(m, 0, 2)
    states = []
    for on_stream in (False, True):
        hook = torch_allreduce(device_buffers=True, on_stream=on_stream)
        cb = _abi.ALLREDUCE_FN(lambda ptr, count, user: hook(ptr, count))
        cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_ls=1.5, item_begin=j0, item_end=j1,
                                  rank=0, world=2,
                                  flags=_abi.ALLREDUCE_ON_STREAM if on_stream else 0)
        cfg.allreduce = cb
        s = Sampler(cfg, y, theta_init=theta)
        hook.stream_ptr = s.stream()
        s.sweep(3)
        states.append((s.state(), s.diagnostics()))
        s.close()
    ok = states[0][1] == states[1][1] and all(
        np.array_equal(states[0][0][k], states[1][0][k]) for k in ("theta", "f", "fstar", "beta"))
    open(os.path.join(out_dir, "r.txt"), "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_stream_ordered_rccl_hook_matches_blocking_hook(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_onstream_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(os.path.join(str(tmp_path), "r.txt")).read() == "ok"


@pytest.mark.parametrize("backend", ["gloo", "host"])
def test_bench_runs_with_two_ranks_under_the_launcher(tmp_path, backend):
    """bench.py exactly as the multi-GPU runs start it (torch.distributed.run, one process per
    rank), rehearsed on this one-GPU box with both ranks on the same device and gloo carrying
    the exchange: the multi-rank plumbing (item ranges, the hook, MAX over ranks, one JSON line
    from rank 0) must work -- with torch's own device tensors ("gloo": torch before the library)
    and with the library's HIP runtime alone ("host": the path the benchmark falls back to when
    no RCCL communicator can be created)."""
    import json
    import subprocess
    env = dict(os.environ, GPIRT_BENCH_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--respondents", "300", "--items", "64", "--no-cpu-baseline"]   # (the launcher's own
    # parser trips over an abbreviated-looking "--n")
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["items_per_gpu"] == 32


def _native_worker(rank, world, port, out_dir):
    """The library's own RCCL collective (libgpirt_rccl.so), no torch on the GPU: (1) the hook on
    a raw device buffer, twice with different lengths (its staging grows); (2) one RCCL rank
    plays rank 0 of 2 -- it owns the first half of the items -- and the stream-ordered native
    hook must drive the chain to the same state as a blocking hook that adds nothing."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    from gpirt_amd import _abi, _lib, comm
    from gpirt_amd.dist import item_range
    from gpirt_amd.sampler import Sampler
    from conftest import load_fixture
    from test_zz_no_alloc import Hip

    _lib.check(_lib.lib().gpirt_set_device(0))
    hip = Hip()
    c = comm.Comm(comm.unique_id(), 0, 1)
    stream = C.c_void_p()
    assert hip.rt.hipStreamCreate(C.byref(stream)) == 0
    c.set_stream(stream.value)
    ok = True
    for count in (1000, 4000, 70):
        x = np.arange(count, dtype=np.float64) * 0.5
        d = hip.upload(x)
        c.allreduce(d.value, count)
        assert hip.rt.hipStreamSynchronize(stream) == 0
        ok = ok and np.array_equal(hip.download(d, (count,)), x)      # one rank: the sum is the vector
        hip.free(d)
    ok = ok and c.calls() == 3

    d = load_fixture("Wiener_C_5")
    y, theta = d["y"][:, :, :3], d["theta"][:, :3]
    n, m, H = y.shape
    j0, j1 = item_range(m, 0, 2)
    states = []
    for native in (False, True):
        cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_ls=1.5, item_begin=j0, item_end=j1,
                                  rank=0, world=2)
        if native:
            c.bind(cfg)
            cfg.world = 2                       # the chain believes in a second rank that adds nothing
        else:
            cb = _abi.ALLREDUCE_FN(lambda ptr, count, user: 0)
            cfg.allreduce = cb
        s = Sampler(cfg, y, theta_init=theta)
        if native:
            c.set_stream(s.stream())
        s.sweep(3)
        states.append((s.state(), s.diagnostics()))
        s.close()
    ok = ok and states[0][1] == states[1][1] and all(
        np.array_equal(states[0][0][k], states[1][0][k]) for k in ("theta", "f", "fstar", "beta"))
    ok = ok and c.calls() > 3
    c.close()
    open(os.path.join(out_dir, "native.txt"), "w").write("ok" if ok else "bad")


def test_native_rccl_collective_without_torch(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_native_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(os.path.join(str(tmp_path), "native.txt")).read() == "ok"


def _fixed_worker(rank, world, port, out_dir):
    """GPIRT_FIXED_ORDER_SUM: `world` ranks share the GPU (host-staged hook); world = 1 runs unsharded."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from gpirt_amd import _abi, _lib
    _lib.check(_lib.lib().gpirt_set_device(0))
    import torch.distributed as dist
    from gpirt_amd.dist import host_staged_allreduce, item_range
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synth import generate

    n, m, C_ = 1536, 2048, 5
    d = generate(n, m, C_, seed=4)
    cfg = _abi.default_config(n=n, m=m, H=1, C=C_, seed=9, flags=_abi.FIXED_ORDER_SUM, rank=rank, world=world)
    keep = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cfg.item_begin, cfg.item_end = item_range(m, rank, world)
        hook = host_staged_allreduce()
        keep = _abi.ALLREDUCE_FN(lambda ptr, count, user: hook(ptr, count))
        cfg.allreduce = keep
    with Sampler(cfg, d["y"], theta_init=d["theta"]) as s:
        s.sweep(3)
        st = s.state(fstar=False)
        seen = s.ll_checksum()
    np.savez(os.path.join(out_dir, "fixed_w%d_r%d.npz" % (world, rank)), j0=cfg.item_begin,
             j1=cfg.item_end if world > 1 else m, ll_checksum=np.uint64(seen),
             **{k: v for k, v in st.items() if v is not None})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_fixed_order_sum_gives_the_same_bits_for_one_two_and_four_ranks(tmp_path):
    """GPIRT_FIXED_ORDER_SUM: the respondent step adds the sums of fixed 512-item blocks in item order
    whatever the sharding, so theta -- and with it every rank's share of f and beta -- is the
    unsharded chain's bit for bit (1536 respondents, 2048 items: 4 blocks; 1, 2 and 4 ranks) -- by
    construction: the log-likelihood sums behind the decisions are equal bit for bit too."""
    import torch.multiprocessing as mp
    out = str(tmp_path)
    for world in (1, 2, 4):
        mp.spawn(_fixed_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    full = np.load(os.path.join(out, "fixed_w1_r0.npz"))
    for world in (2, 4):
        for r in range(world):
            p = np.load(os.path.join(out, "fixed_w%d_r%d.npz" % (world, r)))
            j0, j1 = int(p["j0"]), int(p["j1"])
            assert np.array_equal(p["theta"], full["theta"]), (world, r)
            assert np.array_equal(p["f"], full["f"][:, j0:j1]), (world, r)
            assert np.array_equal(p["beta"], full["beta"][:, j0:j1]), (world, r)
            # the sums themselves, not only the decisions they led to: the checksum covers every
            # log-likelihood sum any decision read, late rounds with few active respondents included
            # (one rank takes the bucketed, block-wise evaluation there too)
            assert int(p["ll_checksum"]) == int(full["ll_checksum"]) != 0, (world, r)


def test_fixed_order_sum_refuses_ranges_off_the_block_grid():
    import ctypes as C
    from gpirt_amd import _abi, _lib
    from gpirt_amd.synth import generate
    n, m = 1536, 2048
    d = generate(n, m, 5, seed=4)
    cb = _abi.ALLREDUCE_FN(lambda ptr, count, user: 0)
    cfg = _abi.default_config(n=n, m=m, H=1, C=5, seed=9, flags=_abi.FIXED_ORDER_SUM, rank=0, world=3,
                              item_begin=0, item_end=683)
    cfg.allreduce = cb
    from gpirt_amd.sampler import Sampler
    with pytest.raises(_lib.GpirtError) as e:
        Sampler(cfg, d["y"], theta_init=d["theta"])
    assert "multiples of 512" in str(e.value)
