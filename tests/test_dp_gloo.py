"""Data-parallel protocol (multimodalgame_amd/dist.py) with world_size 2 over gloo on CPU: two ranks with
half of the minibatch each must end with the parameters one process gets from the whole minibatch
(model.py:1240-1330 on the global batch).  The shard engine is oracle-backed (oracle/dp_ref.py); the GPU
variant of this test (two gloo ranks sharing one MI355X, real HIP engine) is in test_hip_dp.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref, dp_ref
from multimodalgame_amd.dist import DataParallel, shard_range
from tests import common

CASES = {
    "adaptive": dict(use_binary=True, fixed_exchange=False, max_exchange=5, batch_size=12, entropy_s=0.08,
                     entropy_rec=0.01, entropy_sen=0.01, learning_rate=1e-3, top_k_train=2,
                     img_feat_dim=16, img_h_dim=8, rec_w_dim=6, sender_out_dim=6, rec_hidden=5, wv_dim=7, baseline_hid_dim=9),
    "fixed": dict(use_binary=True, fixed_exchange=True, max_exchange=4, batch_size=12, entropy_rec=0.02,
                  learning_rate=1e-3, top_k_train=2,
                  img_feat_dim=16, img_h_dim=8, rec_w_dim=6, sender_out_dim=6, rec_hidden=5, wv_dim=7, baseline_hid_dim=9),
    # -nouse_binary: loss = NLL mean over the global batch -- DataParallel skips the statistics launch and its all-reduce
    "continuous": dict(use_binary=False, fixed_exchange=True, max_exchange=4, batch_size=12, learning_rate=1e-3, top_k_train=2,
                       img_feat_dim=16, img_h_dim=8, rec_w_dim=6, sender_out_dim=6, rec_hidden=5, wv_dim=7, baseline_hid_dim=9),
}
N_CLASSES, N_MB = 4, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _inputs(fl, i):
    x, target, desc = cpu_ref.synthetic_batch(fl.batch_size, N_CLASSES, fl.img_feat_dim, fl.wv_dim, seed=50 + i)
    u = cpu_ref.draw_uniforms(fl.max_exchange, fl.batch_size, fl.rec_w_dim, seed=60 + i)
    return x, target, desc, u


def _worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fl = cpu_ref.Flags(**CASES[case])
    torch.manual_seed(0)
    models = cpu_ref.build_agents(fl)
    cpu_ref.load_filled(models, seed=3)
    eng = dp_ref.ShardEngine(fl, models, global_batch=fl.batch_size)
    dp = DataParallel(eng)
    assert dp.world == world
    lo, n = shard_range(fl.batch_size, rank, world)
    for i in range(N_MB):
        x, target, desc, (u_z, u_s, u_w) = _inputs(fl, i)
        dp.train_step(torch.from_numpy(x[lo:lo + n]), torch.from_numpy(target[lo:lo + n]), torch.from_numpy(desc),
                      u_z[:, lo:lo + n], u_s[:, lo:lo + n], u_w[:, lo:lo + n])
    sd = {"%s.%s" % (a, k): v.numpy() for a, m in models.items() for k, v in m.state_dict().items()}
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **sd)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(CASES))
def test_two_ranks_equal_one_process_on_the_global_batch(case, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    fl = cpu_ref.Flags(**CASES[case])
    torch.manual_seed(0)
    tape = cpu_ref.UniformTape()
    models = cpu_ref.build_agents(fl, rng=tape)
    cpu_ref.load_filled(models, seed=3)
    opts = cpu_ref.build_optimizers(models, fl)
    for i in range(N_MB):
        x, target, desc, (u_z, u_s, u_w) = _inputs(fl, i)
        tape.u = {"z": u_z, "s": u_s, "w": u_w}; tape.t = {"z": 0, "s": 0, "w": 0}
        cpu_ref.train_minibatch(models, opts, torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(desc), fl)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for a, m in models.items():
        for k, v in m.state_dict().items():
            key = "%s.%s" % (a, k)
            np.testing.assert_array_equal(r0[key], r1[key], err_msg="ranks diverged: " + key)
            if key == "receiver.y2.bias":
                continue      # exact gradient is 0; see tests/common.py: compare_packed
            np.testing.assert_allclose(r0[key], v.numpy(), rtol=2e-4, atol=2e-6, err_msg=key)


def test_shard_range():
    assert shard_range(512, 3, 8) == (192, 64)
    with pytest.raises(AssertionError):
        shard_range(10, 0, 4)
