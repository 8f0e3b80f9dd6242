"""World-size-2 rehearsal of the item-sharded chain on CPU (gloo): the partitioning, the
all-reduce hook and the generator's independence of the sharding.  The CPU oracle stands in
for the device library here -- both take the same gpirt_config (item range, world, hook) and
share the lock-step respondent step, so what this covers is the host-side protocol."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, constant_irf, fixed_order=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch.distributed as dist
    from gpirt_amd import _abi
    from gpirt_amd.dist import item_range, torch_allreduce
    from oracle import oracle
    from conftest import load_fixture

    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = load_fixture("Wiener_C_5")
    y, theta = d["y"][:, :, :3], d["theta"][:, :3]
    n, m, H = y.shape
    j0, j1 = item_range(m, rank, world)
    from gpirt_amd.dist import gather_sum_allreduce
    hook = gather_sum_allreduce() if fixed_order else torch_allreduce(device_buffers=False)
    calls = []

    def counted(ptr, count):
        calls.append(count)
        return hook(ptr, count)

    cb = _abi.ALLREDUCE_FN(lambda ptr, count, user: counted(ptr, count))
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_ls=1.5, constant_irf=constant_irf,
                              item_begin=j0, item_end=j1, rank=rank, world=world)
    cfg.allreduce = cb
    s = oracle.Sampler(cfg, y, theta_init=theta)
    s.sweep(3)
    st = s.state()
    diag = s.diagnostics()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), j0=j0, j1=j1, calls=np.array(calls),
             rounds=diag["ess_theta_rounds"], **st)
    s.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("constant_irf", [0, 1])
def test_two_rank_item_sharding_matches_single_rank(tmp_path, constant_irf):
    import torch.multiprocessing as mp
    from gpirt_amd import _abi
    from oracle import oracle
    from conftest import load_fixture

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), constant_irf), nprocs=world, join=True)

    d = load_fixture("Wiener_C_5")
    y, theta = d["y"][:, :, :3], d["theta"][:, :3]
    n, m, H = y.shape
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_ls=1.5, constant_irf=constant_irf)
    ref = oracle.Sampler(cfg, y, theta_init=theta)
    ref.sweep(3)
    full = ref.state()
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert parts[0]["j0"] == 0 and parts[0]["j1"] == parts[1]["j0"] and parts[1]["j1"] == m
    for p in parts:
        j0, j1 = int(p["j0"]), int(p["j1"])
        # theta is replicated: every rank took the same accept/shrink decisions from the
        # all-reduced log-likelihoods; sums are re-associated across ranks, hence not bitwise
        assert np.abs(p["theta"] - full["theta"]).max() < 1e-9
        # the item-local state depends on the sharding only through theta
        assert np.abs(p["f"] - full["f"][:, j0:j1]).max() < 1e-7
        assert np.abs(p["fstar"] - full["fstar"][:, j0:j1]).max() < 1e-7
        assert np.abs(p["beta"] - full["beta"][:, j0:j1]).max() < 1e-9
        fin = np.isfinite(full["threshold"][j0:j1])
        assert np.abs(p["threshold"][fin] - full["threshold"][j0:j1][fin]).max() < 1e-9
        # exactly one n-vector all-reduce per lock-step round
        assert len(p["calls"]) == int(p["rounds"]) and np.all(p["calls"] == n)
    assert np.array_equal(parts[0]["calls"], parts[1]["calls"])


@pytest.mark.parametrize("world", [2, 3, 4])
def test_gather_then_fixed_order_sum_gives_every_rank_the_same_chain(tmp_path, world):
    """The exchange of the native RCCL hook (all-gather + sum in rank order), mirrored on gloo:
    all ranks hold bit-identical theta, and it equals the single-rank chain's theta -- the
    re-associated sums differ at rounding level only, which no accept / shrink comparison of
    these sweeps notices."""
    import torch.multiprocessing as mp
    from gpirt_amd import _abi
    from oracle import oracle
    from conftest import load_fixture

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), 0, True), nprocs=world, join=True)
    d = load_fixture("Wiener_C_5")
    y, theta = d["y"][:, :, :3], d["theta"][:, :3]
    n, m, H = y.shape
    cfg = _abi.default_config(n=n, m=m, H=H, C=5, seed=11, theta_ls=1.5)
    ref = oracle.Sampler(cfg, y, theta_init=theta)
    ref.sweep(3)
    full = ref.state()
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for p in parts:
        j0, j1 = int(p["j0"]), int(p["j1"])
        assert np.array_equal(p["theta"], parts[0]["theta"])          # same bits on every rank
        assert np.array_equal(p["theta"], full["theta"])              # and the unsharded chain's
        assert np.array_equal(p["beta"], full["beta"][:, j0:j1])
        assert np.array_equal(p["f"], full["f"][:, j0:j1])
    assert sum(int(p["j1"]) - int(p["j0"]) for p in parts) == m


def test_item_range_partitions_evenly():
    from gpirt_amd.dist import item_range
    for m in (1, 6, 10, 4096, 4099):
        for world in (1, 2, 3, 4, 8):
            spans = [item_range(m, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
