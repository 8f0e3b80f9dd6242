"""ctypes front-end of ``libgpirt_rccl.so`` (``include/gpirt_comm.h``): the all-reduce of the
item-sharded respondent step on RCCL, with no torch in the data path.  The 128-byte identifier
that rank 0 creates travels to the other ranks over whatever the launcher offers; `exchange_id`
uses the TCP store of ``torch.distributed`` (CPU only, it never touches a GPU) when the job was
started by ``torch.distributed.run``."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgpirt_rccl.so")
ID_BYTES = 128
_lib = None

SYMBOLS = {
    "gpirt_comm_last_error": (C.c_char_p, []),
    "gpirt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gpirt_comm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "gpirt_comm_bind": (C.c_int, [C.c_void_p, C.POINTER(_abi.GpirtConfig)]),
    "gpirt_comm_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gpirt_comm_allreduce": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "gpirt_comm_calls": (C.c_int64, [C.c_void_p]),
    "gpirt_comm_world": (C.c_int, [C.c_void_p]),
    "gpirt_comm_capacity": (C.c_int64, [C.c_void_p]),
    "gpirt_comm_destroy": (None, [C.c_void_p]),
}


class CommError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("gpirt_comm error %d: %s" % (code, message))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("%s is missing: build it with `make -C gpirt_amd/csrc`" % SO_PATH)
        l = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise CommError(rc, lib().gpirt_comm_last_error().decode("utf-8", "replace"))


def unique_id():
    """Rank 0: the communicator's identifier as 128 bytes."""
    buf = C.create_string_buffer(ID_BYTES)
    _check(lib().gpirt_comm_unique_id(buf))
    return buf.raw


def exchange_id(rank, world, key="gpirt_comm_id"):
    """Rank 0 creates the identifier, the others fetch it through the launcher's TCP store
    (MASTER_ADDR / MASTER_PORT of torch.distributed.run); host memory only."""
    if world == 1:
        return unique_id()
    import torch.distributed as dist
    if dist.is_initialized():
        store = dist.distributed_c10d._get_default_store()
    else:
        store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) + 1, world,
                              rank == 0)
    if rank == 0:
        uid = unique_id()
        store.set(key, uid)
        return uid
    return bytes(store.get(key))


def create_all_or_none(rank, world):
    """The communicator on every rank of an initialised torch.distributed (CPU) group, or on none.

    ncclCommInitRank is itself a collective: a rank that cannot even load the library, or rank 0
    failing to create the identifier, would leave the others waiting inside the bootstrap.  So the
    ranks vote BEFORE it -- each loads the library, rank 0 creates the identifier, all-reduce(MIN)
    of the outcome -- and only a unanimous vote goes on; the identifier then travels through the
    same group (no second store, no second port).  A second vote behind the creation lets a rank
    whose RCCL returned an error take every rank to the fallback.  Returns (comm or None, reason)."""
    import torch
    import torch.distributed as dist
    uid, why = None, None
    try:
        lib()
        if rank == 0:
            uid = unique_id()
    except Exception as e:                               # noqa: BLE001
        why = "%s: %s" % (type(e).__name__, e)
    ok = torch.tensor([0 if why else 1])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if not int(ok.item()):
        return None, why or "another rank could not load libgpirt_rccl.so or create the identifier"
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    comm = None
    try:
        comm = Comm(box[0], rank, world)
    except Exception as e:                               # noqa: BLE001
        why = "%s: %s" % (type(e).__name__, e)
    ok = torch.tensor([1 if comm is not None else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if not int(ok.item()):
        if comm is not None:
            comm.close()
        return None, why or "another rank could not create its RCCL communicator"
    return comm, None


class Comm:
    """One rank of the communicator, created on the CURRENT device."""

    def __init__(self, uid, rank, world):
        self._h = C.c_void_p()
        self.rank, self.world = int(rank), int(world)
        _check(lib().gpirt_comm_create(C.create_string_buffer(uid, ID_BYTES), self.rank, self.world,
                                       C.byref(self._h)))

    def bind(self, cfg):
        """rank, world, the all-reduce hook and the stream-ordered flag into a gpirt_config."""
        _check(lib().gpirt_comm_bind(self._h, C.byref(cfg)))

    def set_stream(self, stream_ptr):
        _check(lib().gpirt_comm_set_stream(self._h, C.c_void_p(stream_ptr)))

    def allreduce(self, device_ptr, count):
        _check(lib().gpirt_comm_allreduce(C.c_void_p(device_ptr), int(count), self._h))

    def calls(self):
        return int(lib().gpirt_comm_calls(self._h))

    def world_reported(self):
        """Ranks in the communicator as RCCL counts them (ncclCommCount)."""
        return int(lib().gpirt_comm_world(self._h))

    def capacity(self):
        """Doubles per rank the all-gather's staging holds."""
        return int(lib().gpirt_comm_capacity(self._h))

    def close(self):
        if self._h:
            lib().gpirt_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
