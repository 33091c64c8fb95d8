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


class RcclComm(object):
    """One communicator over all ranks of `group`; all_reduce(tensor) sums in place on the current stream."""

    def __init__(self, device, group=None):
        self.lib = _load()
        self.device = torch.device(device)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        on_gpu = dist.get_backend(group) == "nccl"
        buf = torch.frombuffer(bytearray(bytes(uid.internal)), dtype=torch.uint8).clone()
        if on_gpu:
            buf = buf.to(self.device)
        dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.addressof(uid), bytes(buf.cpu().numpy().tobytes()), 128)
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

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def try_create(device, group=None):
    """Collective: every rank of `group` calls it.  Returns an RcclComm on all ranks or None on all ranks."""
    comm, ok = None, 1
    try:
        comm = RcclComm(device, group)
        # self-check before trusting it with gradients: f32 and f64 sums over the ranks on the current stream
        for dt in (torch.float32, torch.float64):
            probe = torch.full((257,), float(comm.rank + 1), dtype=dt, device=comm.device)
            comm.all_reduce(probe)
            torch.cuda.synchronize(comm.device)
            want = comm.world * (comm.world + 1) / 2.0
            if not bool((probe == want).all().item()):
                raise RuntimeError("direct RCCL all-reduce self-check failed")
    except Exception:                                   # noqa: BLE001 -- any failure means "use torch.distributed"
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        flag = flag.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        if comm is not None:
            comm.close()
        return None
    return comm
