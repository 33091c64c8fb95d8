"""Engine: owns the flat parameter / gradient / optimizer-state buffers and the workspace of one
libmmg handle on one GPU, and exposes the phases of the reference's per-minibatch block
(model.py:1240-1339) as methods.  PyTorch is used for device memory and streams only."""
import ctypes as C

import torch

from . import _lib

_TORCH_DTYPE = {0: torch.float32, 1: torch.uint8, 2: torch.int32, 3: torch.float64}


class Engine(object):
    def __init__(self, device="cuda:0", share=None, **cfg_kwargs):
        """share: another Engine whose flat parameter / gradient / optimizer-state buffers this one
        uses too (same model, different batch size or class count)."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.MmgError("no GPU visible: the exchange path runs on MI355X only (no CPU fallback)")
        self.device = torch.device(device)
        self.cfg = _lib.make_config(**cfg_kwargs)
        self.cfg_kwargs = dict(cfg_kwargs)
        self.use_binary = bool(self.cfg.use_binary)
        n = self.lib.mmg_param_count(C.byref(self.cfg))
        if n < 0:
            raise _lib.MmgError(self.lib.mmg_last_error().decode())
        self.n_params = int(n)
        with torch.cuda.device(self.device):
            if share is not None:
                assert share.n_params == self.n_params and share.device == self.device
                self.flat_params, self.flat_grads, self.opt_state = share.flat_params, share.flat_grads, share.opt_state
            else:
                self.flat_params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
                # gradients + the library's tail quad (dependency-error flag; all-reduced WITH the gradients in DP)
                self.flat_grads = torch.zeros(int(self.lib.mmg_grad_floats(C.byref(self.cfg))), dtype=torch.float32, device=self.device)
                self.opt_state = torch.zeros(2 * self.n_params, dtype=torch.float32, device=self.device)
            ws_bytes = int(self.lib.mmg_workspace_bytes(C.byref(self.cfg)))
            self.workspace = torch.zeros(ws_bytes, dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize(self.device)
            self.handle = self.lib.mmg_create(C.byref(self.cfg), self.workspace.data_ptr(), ws_bytes,
                                              self.flat_params.data_ptr(), self.flat_grads.data_ptr(),
                                              self.opt_state.data_ptr())
        if not self.handle:
            raise _lib.MmgError(self.lib.mmg_last_error().decode())
        # parameter views: {agent: {state_dict key: tensor view into flat_params}}
        self.param_entries = _lib.param_table(self.cfg)
        self.params = {a: {} for a in _lib.AGENTS}
        self.grads = {a: {} for a in _lib.AGENTS}
        for e in self.param_entries:
            numel = e["rows"] * max(e["cols"], 1)
            shape = (e["rows"], e["cols"]) if e["cols"] else (e["rows"],)
            self.params[e["agent"]][e["name"]] = self.flat_params[e["offset"]:e["offset"] + numel].view(shape)
            self.grads[e["agent"]][e["name"]] = self.flat_grads[e["offset"]:e["offset"] + numel].view(shape)
        self.agent_range = {}
        for e in self.param_entries:
            lo, hi = self.agent_range.get(e["agent"], (1 << 62, 0))
            numel = e["rows"] * max(e["cols"], 1)
            self.agent_range[e["agent"]] = (min(lo, e["offset"]), max(hi, e["offset"] + ((numel + 3) // 4) * 4))
        # tape views
        self.tape = {}
        for e in _lib.tape_table(self.cfg):
            dt = _TORCH_DTYPE[e["dtype"]]
            numel = 1
            for d in e["dims"]:
                numel *= d
            nbytes = numel * torch.empty((), dtype=dt).element_size()
            self.tape[e["name"]] = self.workspace[e["offset"]:e["offset"] + nbytes].view(dt).view(e["dims"])
        self.stats = self.tape["stats"]

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.mmg_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t, dtype=None):
        if t is None:
            return None
        assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
        if dtype is not None:
            assert t.dtype == dtype, (t.dtype, dtype)
        return C.c_void_p(t.data_ptr())

    def load_state_dicts(self, sd):
        """sd: {agent: {key: array-like}} -> copied into the flat parameter buffer."""
        for agent, d in sd.items():
            for k, v in d.items():
                self.params[agent][k].copy_(torch.as_tensor(v, dtype=torch.float32).to(self.device).view_as(self.params[agent][k]))

    def state_dicts(self):
        return {a: {k: v.detach().cpu().clone() for k, v in d.items()} for a, d in self.params.items()}

    # ------------------------------------------------------------------ phases
    def forward(self, x, target, desc, u_z=None, u_s=None, u_w=None, seed=0, train=True, run_all=False, minimal=False, log_tape=False):
        """run_all: every sample runs all T steps (what exchange() returns).  minimal (training only): store just what the
        backward pass reads (include/mmg.h: run_all_steps == 2) -- what the fused mmg_train_step does.  log_tape (training
        only): run_all for the conversation, the baselines on the live rows only (run_all_steps == 3: a log minibatch)."""
        f32 = torch.float32
        _lib.check(self.lib.mmg_exchange_forward(
            self.handle, self._ptr(x, f32), self._ptr(target, torch.int64), self._ptr(desc, f32),
            self._ptr(u_z, f32), self._ptr(u_s, f32), self._ptr(u_w, f32), C.c_uint64(seed),
            int(bool(train)), (3 if log_tape and train else 1) if run_all else (2 if minimal and train else 0), self._stream()))

    def loss_stats(self):
        _lib.check(self.lib.mmg_loss_stats(self.handle, self._stream()))

    def backward(self, x, target, desc):
        _lib.check(self.lib.mmg_backward(self.handle, self._ptr(x, torch.float32), self._ptr(target, torch.int64),
                                         self._ptr(desc, torch.float32), self._stream()))

    def clip_step(self):
        _lib.check(self.lib.mmg_clip_step(self.handle, self._stream()))

    def train_step(self, x, target, desc, u_z=None, u_s=None, u_w=None, seed=0):
        f32 = torch.float32
        _lib.check(self.lib.mmg_train_step(
            self.handle, self._ptr(x, f32), self._ptr(target, torch.int64), self._ptr(desc, f32),
            self._ptr(u_z, f32), self._ptr(u_s, f32), self._ptr(u_w, f32), C.c_uint64(seed), self._stream()))

    def train_steps(self, x, target, desc, n, seed=0):
        """n consecutive minibatches enqueued by ONE C call: x [n * B, F], target [n * B] = the epoch's samples in batch order
        (include/mmg.h: mmg_train_steps)."""
        B = self.cfg.batch
        assert x.size(0) >= n * B and target.size(0) >= n * B
        _lib.check(self.lib.mmg_train_steps(self.handle, self._ptr(x, torch.float32), self._ptr(target, torch.int64), int(n),
                                            self._ptr(desc, torch.float32), C.c_uint64(seed), self._stream()))

    def set_allreduce(self, fn_address, comm):
        """RCCL's ncclAllReduce by address + a communicator: the data-parallel step then runs inside ONE C call (dp_train_step)."""
        _lib.check(self.lib.mmg_dp_set_allreduce(self.handle, C.c_void_p(fn_address), comm))

    def dp_train_step(self, x, target, desc, u_z=None, u_s=None, u_w=None, seed=0, full_tape=False, reduce=True):
        f32 = torch.float32
        _lib.check(self.lib.mmg_dp_train_step(
            self.handle, self._ptr(x, f32), self._ptr(target, torch.int64), self._ptr(desc, f32),
            self._ptr(u_z, f32), self._ptr(u_s, f32), self._ptr(u_w, f32), C.c_uint64(seed), int(bool(full_tape)), int(bool(reduce)),
            self._stream()))

    def dp_train_steps(self, x, target, desc, n, seed=0, reduce=True):
        B = self.cfg.batch
        assert x.size(0) >= n * B and target.size(0) >= n * B
        _lib.check(self.lib.mmg_dp_train_steps(self.handle, self._ptr(x, torch.float32), self._ptr(target, torch.int64), int(n),
                                               self._ptr(desc, torch.float32), C.c_uint64(seed), int(bool(reduce)), self._stream()))

    def log_snapshot(self, target, dump=0, losses=True):
        """What the log block of the last minibatch prints, as ONE flat float64 device vector written by one launch
        (include/mmg.h: mmg_log_snapshot)."""
        n = int(self.lib.mmg_log_snapshot_count(C.byref(self.cfg), int(dump), int(bool(losses))))
        out = torch.empty(max(n, 1), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mmg_log_snapshot(self.handle, self._ptr(target, torch.int64) if target is not None else None, int(dump),
                                             int(bool(losses)), C.c_void_p(out.data_ptr()), self._stream()))
        return out[:n]

    def clear_error(self):
        """Clear a recorded in-launch dependency error (drains the stream; the selected kernels stay)."""
        _lib.check(self.lib.mmg_clear_error(self.handle, self._stream()))

    def degraded(self):
        """0: role launches in use; 1: launches without in-launch waits since mmg_create; 2: ... since a recovery."""
        return int(self.lib.mmg_degraded(self.handle))

    # ------------------------------------------------------------------ agent-level steps
    def sender_forward(self, x, w, t, train, u_z=None, seed=0):
        B, W, H = self.cfg.batch, self.cfg.w_dim, self.cfg.h_dim
        msg = torch.empty(B, W, device=self.device)
        probs = torch.empty(B, W, device=self.device) if self.cfg.use_binary else None
        h_x = torch.empty(B, H, device=self.device)
        _lib.check(self.lib.mmg_sender_forward(self.handle, self._ptr(x), self._ptr(w), int(t), int(bool(train)),
                                               self._ptr(u_z), C.c_uint64(seed), self._ptr(msg), self._ptr(probs),
                                               self._ptr(h_x), self._stream()))
        return msg, probs, h_x

    def receiver_forward(self, z, desc, h_z, s_prob_prod, first, t, train, u_s=None, u_w=None, seed=0):
        B, W, R, D = self.cfg.batch, self.cfg.w_dim, self.cfg.rec_hidden, self.cfg.n_classes
        dev = self.device
        s, s_prob = torch.empty(B, 1, device=dev), torch.empty(B, 1, device=dev)
        w = torch.empty(B, W, device=dev)
        w_probs = torch.empty(B, W, device=dev) if self.cfg.use_binary else None
        y, h_w = torch.empty(B, D, device=dev), torch.empty(B, R, device=dev)
        _lib.check(self.lib.mmg_receiver_forward(
            self.handle, self._ptr(z), self._ptr(desc), self._ptr(h_z), self._ptr(s_prob_prod), int(bool(first)),
            int(t), int(bool(train)), self._ptr(u_s), self._ptr(u_w), C.c_uint64(seed), self._ptr(s),
            self._ptr(s_prob), self._ptr(w), self._ptr(w_probs), self._ptr(y), self._ptr(h_w), self._stream()))
        return s, s_prob, w, w_probs, y, h_w

    def baseline_forward(self, which, x, binary, inp):
        rows = binary.shape[0]
        score = torch.empty(rows, 1, device=self.device)
        _lib.check(self.lib.mmg_baseline_forward(self.handle, _lib.AGENTS.index(which), self._ptr(x), self._ptr(binary),
                                                 self._ptr(inp), rows, self._ptr(score), self._stream()))
        return score

    # ------------------------------------------------------------------ profiling
    def set_profiling(self, on):
        _lib.check(self.lib.mmg_set_profiling(self.handle, int(bool(on))))

    def kernel_times(self, max_kernels=512):
        names = C.create_string_buffer(16384)
        ms = (C.c_float * max_kernels)()
        n = self.lib.mmg_get_kernel_times(self.handle, names, 16384, ms, max_kernels)
        if n < 0:
            raise _lib.MmgError(self.lib.mmg_last_error().decode())
        nm = names.value.decode().split(";")[:n]
        return list(zip(nm, [float(ms[i]) for i in range(n)]))

    def check_sync(self):
        """Raise if an in-launch dependency wait (device_utils.h: role_wait) ever hit its spin bound: word 511 of the
        tape array `sync` holds the dependency number + 1.  Synchronises the device; call it off the hot path."""
        code = int(self.tape["sync"][511].item())
        if code:
            raise _lib.MmgError("in-launch dependency %d timed out on the device (workgroup roles out of order?)" % (code - 1))

    # ------------------------------------------------------------------ results
    def losses(self):
        """dict of the six scalars of model.py:1271-1294 plus n_steps / hits (one device->host copy)."""
        v = self.tape["losses"].cpu().tolist()
        self.check_sync()
        keys = ("nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen",
                "n_steps", "hits")
        return dict(zip(keys, v))
