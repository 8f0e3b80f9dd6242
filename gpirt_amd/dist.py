"""Plumbing for one-process-per-GPU runs: item partitioning and the all-reduce hook that
the respondent step calls once per proposal round.  ``torch.distributed`` (backend "nccl",
which is RCCL on ROCm, or "gloo" for CPU rehearsals) carries the collective; nothing else
in the package depends on torch."""
import ctypes as C

import numpy as np


def item_range(m, rank, world):
    """Contiguous, balanced split of items 0..m-1: rank r owns [begin, end)."""
    base, extra = divmod(m, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class _DevVec:
    """Zero-copy view of `count` doubles at a device address for torch.as_tensor."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2,
            "strides": None}


def gather_sum_allreduce(group=None):
    """CPU mirror (gloo, host buffers) of the native hook of ``libgpirt_rccl.so``: all-gather
    the ranks' vectors, then add them in rank order -- every rank performs the same additions
    in the same order, so all ranks end with the same bits."""
    import torch
    import torch.distributed as dist

    def hook(ptr, count):
        arr = np.ctypeslib.as_array((C.c_double * count).from_address(ptr))
        mine = torch.from_numpy(arr.copy())
        parts = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, mine, group=group)
        acc = parts[0].numpy().copy()
        for p in parts[1:]:
            acc += p.numpy()
        arr[:] = acc
        return 0

    return hook


def host_staged_allreduce(group=None):
    """Hook for DEVICE buffers that needs nothing of torch on the GPU: the vector is copied to the
    host with the HIP runtime that ``libgpirt_hip.so`` itself uses, summed as the native hook sums
    it (all-gather over the group, e.g. gloo, then the ranks' vectors added in rank order: the
    same bits on every rank) and copied back.  Blocking (no ``ALLREDUCE_ON_STREAM``): the sampler
    drains its stream before it calls.  For ranks that share one GPU in a test, and as the
    benchmark's fall-back where no RCCL communicator can be created."""
    import torch
    import torch.distributed as dist
    from . import _lib
    copy = _lib.lib().gpirt_device_copy        # the runtime the sampler's buffers live in, whatever torch has loaded

    def hook(ptr, count):
        mine = np.empty(int(count), dtype=np.float64)
        if copy(mine.ctypes.data, C.c_void_p(int(ptr)), 8 * int(count), 0):
            return 1
        parts = [torch.empty(int(count), dtype=torch.float64) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, torch.from_numpy(mine), group=group)
        acc = parts[0].numpy().copy()
        for p in parts[1:]:
            acc += p.numpy()
        return 1 if copy(C.c_void_p(int(ptr)), acc.ctypes.data, 8 * int(count), 1) else 0

    return hook


def torch_allreduce(group=None, device_buffers=True, via_host=False, on_stream=False):
    """Return ``f(ptr, count) -> 0`` summing `count` doubles at `ptr` over the process group,
    in place.  `device_buffers=True`: `ptr` is HBM on the current device (HIP backend);
    False: `ptr` is host memory (CPU rehearsal with gloo).  `via_host=True` stages a device
    buffer through host memory so that a gloo group can carry it (several ranks sharing one
    GPU in a test; RCCL refuses two ranks on one device).

    `on_stream=True` (device buffers, RCCL): the collective is enqueued behind the work
    already on the sampler's HIP stream and the hook returns at once -- set
    ``hook.stream_ptr = sampler.stream()`` after creating the sampler and
    ``ALLREDUCE_ON_STREAM`` in ``cfg.flags`` before.  No host round trip per proposal round."""
    import sys
    from . import _lib
    if device_buffers and "torch" not in sys.modules and _lib.loaded():
        raise RuntimeError(
            "import torch before the first gpirt_amd call that loads libgpirt_hip.so: torch "
            "carries its own copy of the HIP runtime and finds no GPU once the system's copy "
            "has been initialised in the process")
    import torch
    import torch.distributed as dist

    views = {}

    def view(ptr, count):
        key = (int(ptr), int(count))
        if key not in views:
            views[key] = torch.as_tensor(_DevVec(ptr, count), device="cuda")
        return views[key]

    def hook(ptr, count):
        if device_buffers and via_host:
            t = view(ptr, count)
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
            torch.cuda.current_stream().synchronize()
        elif device_buffers and on_stream:
            if hook.stream_ptr is None:
                raise RuntimeError("on_stream hook: set hook.stream_ptr = sampler.stream() first")
            if hook._ext is None:
                hook._ext = torch.cuda.ExternalStream(int(hook.stream_ptr))
            with torch.cuda.stream(hook._ext):
                dist.all_reduce(view(ptr, count), op=dist.ReduceOp.SUM, group=group)
        elif device_buffers:
            dist.all_reduce(view(ptr, count), op=dist.ReduceOp.SUM, group=group)
            torch.cuda.current_stream().synchronize()
        else:
            arr = np.ctypeslib.as_array((C.c_double * count).from_address(ptr))
            t = torch.from_numpy(arr)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0

    hook.stream_ptr = None
    hook._ext = None
    return hook
