"""ctypes binding of libmmg.so (include/mmg.h).  The product path has NO fallback: if the HIP
library is missing or a call fails, an exception is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmg.so")

MMG_OPT = {"RMSprop": 0, "Adam": 1, "SGD": 2}
AGENTS = ("receiver", "sender", "baseline_rec", "baseline_sen")      # MMG_AGENT_* order


class MmgConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "global_batch", "batch_offset", "n_classes", "feat_dim", "h_dim", "w_dim", "rec_hidden",
        "wv_dim", "bas_hidden", "max_exchange", "use_binary", "fixed_exchange", "s_prob_prod",
        "has_entropy_s", "has_entropy_sen", "has_entropy_rec")] + [
        ("entropy_s", C.c_float), ("entropy_sen", C.c_float), ("entropy_rec", C.c_float),
        ("first_rec", C.c_float), ("optim_type", C.c_int32), ("learning_rate", C.c_float), ("top_k", C.c_int32),
        ("cu_budget", C.c_int32)]


class ParamEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("agent", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("offset", C.c_int64)]


class TapeEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("dtype", C.c_int32), ("ndim", C.c_int32), ("dims", C.c_int64 * 4),
                ("offset", C.c_int64)]


class MmgError(RuntimeError):
    pass


class MmgWarning(RuntimeWarning):
    """A training call returned 1: the library recovered from a timed-out in-launch dependency (include/mmg.h: fail-soft)."""


_lib = None

# every symbol include/mmg.h declares (tests check that the library exports all of them)
SYMBOLS = ["mmg_last_error", "mmg_version", "mmg_param_count", "mmg_grad_floats", "mmg_param_table", "mmg_workspace_bytes",
           "mmg_tape_table", "mmg_create", "mmg_destroy", "mmg_exchange_forward", "mmg_loss_stats",
           "mmg_backward", "mmg_clip_step", "mmg_train_step", "mmg_sender_forward", "mmg_receiver_forward",
           "mmg_baseline_forward", "mmg_set_profiling", "mmg_get_kernel_times", "mmg_host_shuffle", "mmg_train_steps",
           "mmg_dp_set_allreduce", "mmg_dp_train_step", "mmg_dp_train_steps", "mmg_clear_error", "mmg_degraded",
           "mmg_log_snapshot_count", "mmg_log_snapshot"]


def load():
    """Load libmmg.so; raises MmgError when it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmgError("HIP library %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, fp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_void_p
    cfgp = C.POINTER(MmgConfig)
    lib.mmg_last_error.restype = C.c_char_p
    lib.mmg_version.restype = i32
    lib.mmg_param_count.restype = i64; lib.mmg_param_count.argtypes = [cfgp]
    lib.mmg_grad_floats.restype = i64; lib.mmg_grad_floats.argtypes = [cfgp]
    lib.mmg_param_table.restype = i32; lib.mmg_param_table.argtypes = [cfgp, C.POINTER(ParamEntry), i32]
    lib.mmg_workspace_bytes.restype = i64; lib.mmg_workspace_bytes.argtypes = [cfgp]
    lib.mmg_tape_table.restype = i32; lib.mmg_tape_table.argtypes = [cfgp, C.POINTER(TapeEntry), i32]
    lib.mmg_create.restype = vp; lib.mmg_create.argtypes = [cfgp, vp, i64, fp, fp, fp]
    lib.mmg_destroy.restype = None; lib.mmg_destroy.argtypes = [vp]
    lib.mmg_exchange_forward.restype = i32
    lib.mmg_exchange_forward.argtypes = [vp, fp, vp, fp, fp, fp, fp, u64, i32, i32, vp]
    lib.mmg_loss_stats.restype = i32; lib.mmg_loss_stats.argtypes = [vp, vp]
    lib.mmg_backward.restype = i32; lib.mmg_backward.argtypes = [vp, fp, vp, fp, vp]
    lib.mmg_clip_step.restype = i32; lib.mmg_clip_step.argtypes = [vp, vp]
    lib.mmg_train_step.restype = i32; lib.mmg_train_step.argtypes = [vp, fp, vp, fp, fp, fp, fp, u64, vp]
    lib.mmg_train_steps.restype = i32; lib.mmg_train_steps.argtypes = [vp, fp, vp, i64, fp, u64, vp]
    lib.mmg_dp_set_allreduce.restype = i32; lib.mmg_dp_set_allreduce.argtypes = [vp, vp, vp]
    lib.mmg_dp_train_step.restype = i32; lib.mmg_dp_train_step.argtypes = [vp, fp, vp, fp, fp, fp, fp, u64, i32, i32, vp]
    lib.mmg_dp_train_steps.restype = i32; lib.mmg_dp_train_steps.argtypes = [vp, fp, vp, i64, fp, u64, i32, vp]
    lib.mmg_clear_error.restype = i32; lib.mmg_clear_error.argtypes = [vp, vp]
    lib.mmg_degraded.restype = i32; lib.mmg_degraded.argtypes = [vp]
    lib.mmg_log_snapshot_count.restype = i64; lib.mmg_log_snapshot_count.argtypes = [cfgp, i32, i32]
    lib.mmg_log_snapshot.restype = i32; lib.mmg_log_snapshot.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.mmg_sender_forward.restype = i32
    lib.mmg_sender_forward.argtypes = [vp, fp, fp, i32, i32, fp, u64, fp, fp, fp, vp]
    lib.mmg_receiver_forward.restype = i32
    lib.mmg_receiver_forward.argtypes = [vp, fp, fp, fp, fp, i32, i32, i32, fp, fp, u64, fp, fp, fp, fp, fp, fp, vp]
    lib.mmg_baseline_forward.restype = i32; lib.mmg_baseline_forward.argtypes = [vp, i32, fp, fp, fp, i32, fp, vp]
    lib.mmg_host_shuffle.restype = i32; lib.mmg_host_shuffle.argtypes = [vp, i32, i64, vp]
    lib.mmg_set_profiling.restype = i32; lib.mmg_set_profiling.argtypes = [vp, i32]
    lib.mmg_get_kernel_times.restype = i32
    lib.mmg_get_kernel_times.argtypes = [vp, C.c_char_p, i32, C.POINTER(C.c_float), i32]
    _lib = lib
    return lib


def check(status):
    """0 = ok; 1 = ok with a warning (the library recovered from a timed-out in-launch dependency and continues on the launches
    without in-launch waits: reported once per event as MmgWarning); negative = error."""
    if status == 0:
        return
    if status > 0:
        import warnings
        warnings.warn(MmgWarning(load().mmg_last_error().decode()), stacklevel=3)
        return
    raise MmgError(load().mmg_last_error().decode())


def make_config(batch, n_classes, feat_dim, h_dim, w_dim, rec_hidden, wv_dim, bas_hidden, max_exchange,
                use_binary=True, fixed_exchange=True, s_prob_prod=True, entropy_s=None, entropy_sen=None,
                entropy_rec=None, first_rec=0.0, optim_type="RMSprop", learning_rate=1e-4, top_k=6,
                global_batch=None, batch_offset=0, cu_budget=0):
    """cu_budget: compute units this process can count on (0 = the whole device; include/mmg.h)."""
    c = MmgConfig()
    c.batch, c.global_batch, c.batch_offset = batch, global_batch or batch, batch_offset
    c.n_classes, c.feat_dim, c.h_dim, c.w_dim = n_classes, feat_dim, h_dim, w_dim
    c.rec_hidden, c.wv_dim, c.bas_hidden, c.max_exchange = rec_hidden, wv_dim, bas_hidden, max_exchange
    c.use_binary, c.fixed_exchange, c.s_prob_prod = int(bool(use_binary)), int(bool(fixed_exchange)), int(bool(s_prob_prod))
    c.has_entropy_s, c.has_entropy_sen, c.has_entropy_rec = [int(v is not None) for v in (entropy_s, entropy_sen, entropy_rec)]
    c.entropy_s, c.entropy_sen, c.entropy_rec = [float(v or 0.0) for v in (entropy_s, entropy_sen, entropy_rec)]
    c.first_rec, c.optim_type, c.learning_rate, c.top_k = float(first_rec), MMG_OPT[optim_type], float(learning_rate), int(top_k)
    c.cu_budget = int(cu_budget or 0)
    return c


def param_table(cfg):
    lib = load()
    n = lib.mmg_param_table(C.byref(cfg), None, 0)
    if n < 0:
        raise MmgError(lib.mmg_last_error().decode())
    arr = (ParamEntry * n)()
    check(0 if lib.mmg_param_table(C.byref(cfg), arr, n) == n else -1)
    return [dict(name=e.name.decode(), agent=AGENTS[e.agent], rows=e.rows, cols=e.cols, offset=e.offset) for e in arr]


def tape_table(cfg):
    lib = load()
    n = lib.mmg_tape_table(C.byref(cfg), None, 0)
    if n < 0:
        raise MmgError(lib.mmg_last_error().decode())
    arr = (TapeEntry * n)()
    check(0 if lib.mmg_tape_table(C.byref(cfg), arr, n) == n else -1)
    return [dict(name=e.name.decode(), dtype=e.dtype, dims=[int(e.dims[i]) for i in range(e.ndim)], offset=e.offset)
            for e in arr]
