"""Data-parallel path with the REAL HIP engine: two gloo ranks share the single MI355X of the test box,
each owns half of the minibatch (Engine(batch=B/2, global_batch=B, batch_offset=rank*B/2)) and runs
multimodalgame_amd.dist.DataParallel.train_step; the result must equal one process with the whole batch.
Also checks that in-kernel Philox sampling is invariant to the sharding (keyed by the GLOBAL sample index)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import common

pytestmark = pytest.mark.gpu
NAME = "g2_adaptive_c1"


def _case(case):
    """(name, meta): a golden case, or "c5": BASELINE configs[4]'s agents (D = 1000, continuous, Fixed) at 64 samples -- the
    many-class path (k_conversation_mc / k_bwd_mc1/2), whose data-parallel step has no statistics collective."""
    if case == "c5":
        from oracle import cpu_ref
        fl = cpu_ref.Flags(use_binary=False, fixed_exchange=True, max_exchange=10, batch_size=64, learning_rate=1e-4,
                           img_feat_dim=512, img_h_dim=256, rec_w_dim=32, sender_out_dim=32, rec_hidden=64, wv_dim=100,
                           baseline_hid_dim=500, top_k_train=6)
        meta = dict(fl.__dict__)
        meta.update(n_classes=1000, batch=64, n_minibatches=2, seed_weights=5, seed_data=6, seed_uniforms=7)
        return None, meta
    return case, common.load_golden(case)[1]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(eng, dp, meta, lo, n, philox, name=NAME):
    dev = eng.device
    for i in range(meta["n_minibatches"]):
        x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, i, name)
        xd, td, dd = [torch.from_numpy(a).to(dev) for a in (x[lo:lo + n], target[lo:lo + n], desc)]
        if philox:
            u = (None, None, None)
        else:
            u = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (u_z[:, lo:lo + n], u_s[:, lo:lo + n, 0], u_w[:, lo:lo + n])]
        if dp is None:
            eng.train_step(xd, td, dd, *u, seed=1234)
        else:
            dp.train_step(xd, td, dd, *u, seed=1234)
    torch.cuda.synchronize()
    out = {"%s.%s" % (a, k): v.cpu().numpy() for a, d in eng.params.items() for k, v in d.items()}
    out["losses"] = eng.tape["losses"].cpu().numpy()
    out["totals"] = eng.tape["totals"].cpu().numpy()
    return out


def _worker(rank, world, port, philox, out_dir, case=NAME):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodalgame_amd.dist import DataParallel, shard_range
    name, meta = _case(case)
    lo, n = shard_range(meta["batch"], rank, world)
    eng = common.make_engine(meta, batch=n, global_batch=meta["batch"], batch_offset=lo)
    out = _run(eng, DataParallel(eng), meta, lo, n, philox, name)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,philox", [(NAME, False), (NAME, True), ("g3_continuous", False), ("c5", True)])
def test_two_ranks_on_one_gpu_equal_single_process(case, philox, tmp_path):
    """Binary / Adaptive: statistics all-reduce + gradient all-reduce.  Continuous cases ("g3_continuous": register-resident
    kernels, k_stats launched by mmg_backward itself; "c5": many-class path, statistics workgroup inside k_bwd_mc2): ONE
    collective -- the logged NLL and hit count of the global minibatch come out of the gradient buffer's tail quad."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), philox, str(tmp_path), case), nprocs=world, join=True)
    name, meta = _case(case)
    eng = common.make_engine(meta)
    want = _run(eng, None, meta, 0, meta["batch"], philox, name)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k, v in want.items():
        np.testing.assert_array_equal(r0[k], r1[k], err_msg="ranks diverged: " + k)
        if k == "receiver.y2.bias":
            continue
        np.testing.assert_allclose(r0[k], v, rtol=2e-4, atol=2e-6, err_msg=k)


def _rccl_worker(rank, world, port, direct, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)          # "nccl" is RCCL on ROCm
    from multimodalgame_amd.dist import DataParallel
    z, meta = common.load_golden(NAME)
    eng = common.make_engine(meta)
    dp = DataParallel(eng, direct=direct)
    assert (dp.comm is not None) == direct
    dp.world = 2            # force both collectives to run even though the group has one member
    out = _run(eng, dp, meta, 0, meta["batch"], True)
    np.savez(os.path.join(out_dir, "rccl.npz"), **out)
    eng2 = common.make_engine(meta)
    dp2 = DataParallel(eng2)
    dp2.world = 1           # the same split call sequence without the collectives
    np.savez(os.path.join(out_dir, "split.npz"), **_run(eng2, dp2, meta, 0, meta["batch"], True))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("direct", [False, True], ids=["torch.distributed", "direct-rccl"])
def test_rccl_collectives_on_engine_buffers(direct, tmp_path):
    """direct=True: multimodalgame_amd.rccl (ncclAllReduce on the engine's stream); False: torch.distributed.
    The production backend: RCCL all-reduces (f64 statistics vector, f32 flat gradient buffer) issued on the
    engine's own device buffers between the C-ABI calls.  One box = one GPU, so the group has a single member:
    the sums are identities and the result must equal the same call sequence without collectives bit for bit
    (and the fused mmg_train_step within rounding: it takes the gradient norm from k_wgrad's partials)."""
    mp.spawn(_rccl_worker, args=(1, _free_port(), direct, str(tmp_path)), nprocs=1, join=True)
    z, meta = common.load_golden(NAME)
    fused = _run(common.make_engine(meta), None, meta, 0, meta["batch"], True)
    got, want = np.load(tmp_path / "rccl.npz"), np.load(tmp_path / "split.npz")
    for k in want.files:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        if k != "receiver.y2.bias":
            np.testing.assert_allclose(got[k], fused[k], rtol=2e-4, atol=2e-6, err_msg=k)


def _rccl2_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from multimodalgame_amd import rccl
    from multimodalgame_amd.dist import DataParallel
    dev = torch.device("cuda", rank)
    comm = rccl.try_create(dev)                      # ncclCommInitRank with the by-value 128-byte id through ctypes, world = 2
    assert comm is not None and comm.world == 2
    for dt in (torch.float32, torch.float64):
        t = torch.arange(1000, dtype=dt, device=dev) * (rank + 1)
        comm.all_reduce(t)
        torch.cuda.synchronize(dev)
        assert bool((t == torch.arange(1000, dtype=dt, device=dev) * 3).all())
    comm.close()
    z, meta = common.load_golden(NAME)
    B = meta["batch"] // world
    eng = common.make_engine(meta, batch=B, global_batch=meta["batch"], batch_offset=rank * B, device=dev)
    dp = DataParallel(eng)
    assert dp.comm is not None
    out = _run(eng, dp, meta, rank * B, B, True)
    np.savez(os.path.join(out_dir, "nccl_rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: multi-rank RCCL (ncclCommInitRank with world > 1, "
                    "f32 / f64 ncclAllReduce over xGMI) cannot execute on a one-GPU box")
def test_two_rank_rccl_data_parallel(tmp_path):
    """World size 2 on two GPUs through multimodalgame_amd.rccl: the direct communicator's all-reduces and a sharded
    training step equal to the single-GPU step on the whole batch."""
    world = 2
    mp.spawn(_rccl2_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z, meta = common.load_golden(NAME)
    want = _run(common.make_engine(meta), None, meta, 0, meta["batch"], True)
    r0, r1 = np.load(tmp_path / "nccl_rank0.npz"), np.load(tmp_path / "nccl_rank1.npz")
    for k, v in want.items():
        np.testing.assert_array_equal(r0[k], r1[k], err_msg="ranks diverged: " + k)
        if k != "receiver.y2.bias":
            np.testing.assert_allclose(r0[k], v, rtol=2e-4, atol=2e-6, err_msg=k)
