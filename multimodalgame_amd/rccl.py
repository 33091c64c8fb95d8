"""RCCL communicator driven directly (ctypes on the librccl.so PyTorch-ROCm already loaded), so that the two
all-reduces of a data-parallel step are enqueued on the SAME HIP stream as the engine's kernels: no cross-stream
event record/wait per collective and no c10d dispatch.  On one MI355X a single-member torch.distributed "nccl"
all-reduce costs ~13 us of stream hand-over around a no-op; the step itself is ~100 us, so two of them matter.

torch.distributed is still what launches the job (one process per GPU, torch.distributed.run) and is used
once, to broadcast the 128-byte ncclUniqueId.  If anything in the direct set-up fails on any rank, every rank
falls back to torch.distributed.all_reduce (which is RCCL as well)."""
import ctypes as C
import os

import torch
import torch.distributed as dist

_NCCL_SUM = 0
_DTYPES = {torch.float32: 7, torch.float64: 8}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path)
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    return lib


def _agree(ok, device, group):
    """Collective MIN of a per-rank success flag over torch.distributed: every rank learns whether ALL succeeded."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        flag = flag.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


class RcclComm(object):
    """One communicator over all ranks of `group`; all_reduce(tensor) sums in place on the current stream.
    Construct through try_create(): the set-up is a sequence of collective stages with an agreement after each."""

    def __init__(self, lib, device, rank, world, uid):
        self.lib, self.device, self.rank, self.world = lib, torch.device(device), rank, world
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_reduce(self, tensor):
        assert tensor.is_contiguous() and tensor.device == self.device
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.ncclAllReduce(C.c_void_p(tensor.data_ptr()), C.c_void_p(tensor.data_ptr()), tensor.numel(),
                                           _DTYPES[tensor.dtype], _NCCL_SUM, self.comm, C.c_void_p(stream)), "ncclAllReduce")

    def all_reduce_address(self):
        """Address of RCCL's ncclAllReduce: libmmg calls it on this communicator from inside mmg_dp_train_step."""
        return C.cast(self.lib.ncclAllReduce, C.c_void_p).value

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def try_create(device, group=None):
    """Collective: every rank of `group` calls it.  Returns an RcclComm on ALL ranks or None on ALL ranks.

    Every stage that can fail locally is followed by an agreement (MIN all-reduce of a success flag through
    torch.distributed), and a rank only enters the next RCCL call when all ranks passed the previous stage -- so a rank-local
    failure (library not loadable, ncclGetUniqueId error, init error reported by RCCL, wrong self-check sum) never leaves
    the other ranks blocked in a collective this rank will not join:
      1. load librccl + (rank 0) ncclGetUniqueId           -> agree
      2. broadcast of the 128-byte id (always executed)     -> ncclCommInitRank on every rank -> agree
      3. f32 AND f64 self-check all-reduces, both unconditionally, result compared afterwards -> agree
    Residual risk: ncclCommInitRank itself is a blocking rendezvous -- if a rank dies inside it the others wait for RCCL's own
    bootstrap timeout (NCCL_SOCKET / bootstrap settings); MMG_DP_DIRECT_RCCL=0 selects torch.distributed outright."""
    device = torch.device(device)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lib, uid, ok = None, _UniqueId(), True
    try:
        lib = _load()
        if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
            ok = False
    except Exception:                                   # noqa: BLE001
        ok = False
    if not _agree(ok, device, group):
        return None
    buf = torch.frombuffer(bytearray(bytes(uid.internal)), dtype=torch.uint8).clone()
    if dist.get_backend(group) == "nccl":
        buf = buf.to(device)
    dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    C.memmove(C.addressof(uid), bytes(buf.cpu().numpy().tobytes()), 128)
    comm = None
    try:
        comm = RcclComm(lib, device, rank, world, uid)
    except Exception:                                   # noqa: BLE001
        ok = False
    if not _agree(ok, device, group):
        if comm is not None:
            comm.close()
        return None
    # self-check before trusting it with gradients: both probes run on every rank, verdict afterwards
    want = world * (world + 1) / 2.0
    try:
        probes = [torch.full((257,), float(rank + 1), dtype=dt, device=device) for dt in (torch.float32, torch.float64)]
        for p in probes:
            comm.all_reduce(p)
        torch.cuda.synchronize(device)
        ok = all(bool((p == want).all().item()) for p in probes)
    except Exception:                                   # noqa: BLE001
        ok = False
    if not _agree(ok, device, group):
        comm.close()
        return None
    return comm
