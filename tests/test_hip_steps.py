"""The C-driven loops of round 6 (include/mmg.h): mmg_train_steps (n minibatches of the epoch loop, model.py:1218-1240, enqueued by
ONE call) and mmg_dp_train_step (the data-parallel minibatch in ONE call, the collectives being RCCL's ncclAllReduce called by
address on the engine's stream) must take, bit for bit, the steps of the per-minibatch / phased call sequences they replace."""
import os

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu


def _epoch(meta, n, dev, seed=0):
    rs = np.random.RandomState(seed)
    B, F, D = int(meta["batch"]), int(meta["img_feat_dim"]), int(meta["n_classes"])
    x = np.abs(rs.standard_normal((n * B, F))).astype(np.float32)
    t = rs.randint(0, D, size=(n * B,)).astype(np.int64)
    desc = (0.3 * rs.standard_normal((D, int(meta["wv_dim"])))).astype(np.float32)
    return [torch.from_numpy(a).to(dev) for a in (x, t, desc)]


@pytest.mark.parametrize("name", ["g2_adaptive_c1", "g3_fixed_c3shard", "g3_continuous"])
def test_train_steps_equals_n_train_step_calls(name):
    """Same parameters, optimizer state, sampling counters and logged totals after 7 minibatches: one mmg_train_steps call against
    seven mmg_train_step calls on consecutive [B, F] slices of the batch-ordered epoch (misc.Epoch)."""
    z, meta = common.load_golden(name)
    a, b = common.make_engine(meta), common.make_engine(meta)
    n, B = 7, int(meta["batch"])
    x, t, desc = _epoch(meta, n, a.device)
    a.train_steps(x, t, desc, n, seed=11)
    for i in range(n):
        b.train_step(x[i * B:(i + 1) * B], t[i * B:(i + 1) * B], desc, seed=11)
    torch.cuda.synchronize()
    a.check_sync(); b.check_sync()
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.opt_state, b.opt_state)
    assert torch.equal(a.tape["counter"], b.tape["counter"]) and int(a.tape["counter"][0]) == n
    assert torch.equal(a.tape["totals"], b.tape["totals"]) and torch.equal(a.tape["losses"], b.tape["losses"])


@pytest.mark.parametrize("name", ["g2_adaptive_c1", "g3_continuous"])
def test_dp_train_step_in_the_library_equals_the_phased_calls(name):
    """One rank of a two-rank job (global_batch = 2 B) WITHOUT collectives (reduce = 0): mmg_dp_train_step == forward |
    loss_stats | backward | clip_step, bit for bit, for plain and for full-tape (log) minibatches."""
    z, meta = common.load_golden(name)
    B = int(meta["batch"])
    a, b = [common.make_engine(meta, global_batch=2 * B) for _ in range(2)]
    x, t, desc = _epoch(meta, 4, a.device, seed=3)
    for i in range(4):
        xs, ts = x[i * B:(i + 1) * B], t[i * B:(i + 1) * B]
        full = i == 2
        a.dp_train_step(xs, ts, desc, seed=5, full_tape=full, reduce=False)
        b.forward(xs, ts, desc, seed=5, train=True, run_all=full, minimal=not full, log_tape=True)
        if b.use_binary:
            b.loss_stats()
        b.backward(xs, ts, desc)
        b.clip_step()
    torch.cuda.synchronize()
    a.check_sync(); b.check_sync()
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.opt_state, b.opt_state)
    assert torch.equal(a.flat_grads, b.flat_grads) and torch.equal(a.tape["losses"], b.tape["losses"])


def _rccl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from multimodalgame_amd.dist import DataParallel
        z, meta = common.load_golden("g2_adaptive_c1")
        a, b = common.make_engine(meta), common.make_engine(meta)
        dp = DataParallel(a, direct=True)
        assert dp.comm is not None and dp.in_library, "direct RCCL communicator not available"
        B = int(meta["batch"])
        x, t, desc = _epoch(meta, 5, a.device, seed=9)
        # the collectives of a one-member communicator are identities: the step must equal the phased sequence exactly, with the
        # two ncclAllReduce calls (f64 statistics, f32 gradients + tail quad) enqueued from inside the library
        a.dp_train_step(x[:B], t[:B], desc, seed=2, reduce=True)
        a.dp_train_steps(x[B:], t[B:], desc, 4, seed=2, reduce=True)
        for i in range(5):
            xs, ts = x[i * B:(i + 1) * B], t[i * B:(i + 1) * B]
            b.forward(xs, ts, desc, seed=2, train=True, run_all=False, minimal=True)
            b.loss_stats(); b.backward(xs, ts, desc); b.clip_step()
        torch.cuda.synchronize()
        a.check_sync()
        ok = torch.equal(a.flat_params, b.flat_params) and torch.equal(a.opt_state, b.opt_state)
        open(out, "w").write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_dp_train_step_calls_rccl_by_address_on_the_engine_stream(tmp_path):
    """mmg_dp_set_allreduce(ncclAllReduce, comm) + mmg_dp_train_step[s](reduce = 1) on a ONE-rank RCCL communicator (the box has
    one GPU): exercises the function-pointer call into librccl from libmmg (datatype / op codes, in-place buffers, stream)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_rccl_worker, args=(1, port, out), nprocs=1, join=True)
    assert open(out).read() == "ok"


@pytest.mark.parametrize("name,dump", [("g2_adaptive_c1", 3), ("g2_adaptive_c1", 0), ("g3_fixed_c3shard", 2), ("g3_continuous", 0)])
def test_log_snapshot_kernel_equals_the_torch_form(name, dump, monkeypatch):
    """mmg_log_snapshot (ONE launch: the log block's losses, statistics, per-step prediction entropies over the whole batch,
    predictions and sample-dump slices as one flat f64 vector, model.py:1342-1461) against the ~16 torch ops model.run() used
    through round 5 (MMG_LOG_TORCH=1) on the same run-all tape: copied values exact, entropies within float32 rounding."""
    from multimodalgame_amd import model
    z, meta = common.load_golden(name)
    eng = common.make_engine(meta)
    x, t, desc = _epoch(meta, 1, eng.device, seed=4)
    eng.forward(x, t, desc, seed=3, train=True, run_all=True)
    if eng.use_binary:
        eng.loss_stats()
    eng.backward(x, t, desc)
    eng.clip_step()
    a = model._log_snapshot_end(model._log_snapshot_begin(eng, t, dump=dump))
    b_eval = model._log_snapshot_end(model._log_snapshot_begin(eng, None, dump=max(dump, 1), losses=False))
    monkeypatch.setenv("MMG_LOG_TORCH", "1")
    want = model._log_snapshot_end(model._log_snapshot_begin(eng, t, dump=dump))
    want_eval = model._log_snapshot_end(model._log_snapshot_begin(eng, None, dump=max(dump, 1), losses=False))
    assert a["losses"] == want["losses"] and a["hits_total"] == want["hits_total"] and a["stats"] == want["stats"]
    assert a["argmax"] == want["argmax"] and a["target"] == want["target"]
    np.testing.assert_allclose(a["ent_y"], want["ent_y"], rtol=2e-5, atol=2e-6)
    assert ("dump" in a) == (dump > 0)
    for got, ref in ((a, want), (b_eval, want_eval)):
        if "dump" in ref:
            assert got["dump"] == ref["dump"]
