"""The data-parallel EPOCH LOOP of model.run() (reference loop model.py:1190-1240, batch order
misc.py:257-302) on the CPU: two gloo ranks of `python -m multimodalgame_amd.model` on a synthetic HDF5, each keeping its half
of every minibatch, must end with the parameters of the one-process run -- same batches, same sampled bits, same update.
The engine under the loop is the oracle-backed stand-in of tests/oracle_engine.py (swapped in HERE, explicitly: the product has
no CPU path); the GPU variant with the real HIP engine is tests/test_cli_gpu.py::test_data_parallel_model_run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

N_CLASSES, PER_CLASS, BATCH = 6, 8, 12          # 48 samples -> 4 minibatches per epoch


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _argv(tmp, name, mode, extra=()):
    data = os.path.join(tmp, "data")
    argv = ["model.py", "-experiment_name", name, "-log_path", os.path.join(tmp, "logs_" + name), "-model_type", mode[0],
            "-batch_size", str(BATCH), "-batch_size_dev", "16", "-max_exchange", "4", "-rec_w_dim", "6", "-sender_out_dim", "6",
            "-img_h_dim", "8", "-img_feat_dim", "16", "-rec_hidden", "5", "-wv_dim", "7", "-baseline_hid_dim", "9",
            "-learning_rate", "1e-3", "-top_k_train", "2", "-top_k_dev", "2", "-max_epoch", "2", "-log_interval", "3",
            "-log_dev", "5", "-save_after", "0", "-save_interval", "1", "-exchange_samples", "0", "-seed", "3",
            "-dist_backend", "gloo"] + list(mode[1]) + list(extra)
    for k in ("train_file", "dev_file", "descr_train", "descr_dev", "glove_path"):
        argv += ["-" + k, PATHS[k]]
    return argv


PATHS = {}
MODES = {"adaptive": ("Adaptive", ["-use_binary", "-entropy_s", "0.08", "-entropy_rec", "0.01", "-entropy_sen", "0.01"]),
         "continuous": ("Fixed", ["-nouse_binary"])}


def _run_model(argv, out_npz):
    from multimodalgame_amd import flags as _flags, game as _game, model as _model
    from tests import oracle_engine
    _game.Engine = oracle_engine.OracleEngine                      # the swap: oracle math under the product's host code
    _model._device = lambda local_rank: torch.device("cpu")
    oracle_engine.OracleEngine.instances = []
    _flags.define_flags()
    _flags.FLAGS.Reset()
    _flags.FLAGS(argv)
    _flags.FLAGS.img_feat_dim = 16                                  # (the presets set 512: the synthetic file has 16 features)
    _flags.default_flags(argv)
    _flags.FLAGS.img_feat_dim = 16
    stats = {}
    _model.run(stats=stats)
    eng = oracle_engine.OracleEngine.instances[0]
    np.savez(out_npz, params=eng.flat_params.numpy(), totals=eng.tape["totals"].numpy(), minibatches=stats["minibatches"],
             exchange_steps=stats["exchange_steps"], y2_bias=eng.params["receiver"]["y2.bias"].storage_offset())


def _worker(rank, world, port, tmp, paths, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    PATHS.update(paths)
    _run_model(_argv(tmp, "dp", MODES[mode]), os.path.join(tmp, "dp_rank%d.npz" % rank))


def _single(tmp, paths, mode):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    torch.set_num_threads(1)
    PATHS.update(paths)
    _run_model(_argv(tmp, "one", MODES[mode]), os.path.join(tmp, "one.npz"))


@pytest.mark.parametrize("mode", sorted(MODES))
def test_two_rank_model_run_equals_one_process(mode, tmp_path):
    from multimodalgame_amd import misc
    tmp = str(tmp_path)
    paths = misc.write_synthetic_dataset(os.path.join(tmp, "data"), n_classes=N_CLASSES, per_class=PER_CLASS, feat_dim=16, wv_dim=7)
    mp.spawn(_worker, args=(2, _free_port(), tmp, paths, mode), nprocs=2, join=True)
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_single, args=(tmp, paths, mode)); p.start(); p.join()
    assert p.exitcode == 0
    r0, r1, one = (np.load(os.path.join(tmp, f)) for f in ("dp_rank0.npz", "dp_rank1.npz", "one.npz"))
    assert int(one["minibatches"]) == int(r0["minibatches"]) == 8                  # 2 epochs x 4 minibatches, none dropped twice
    np.testing.assert_array_equal(r0["params"], r1["params"])                       # the ranks never diverge
    keep = np.arange(r0["params"].size) != int(r0["y2_bias"])     # y2.bias: exact gradient 0, RMSprop amplifies rounding noise (tests/common.py)
    np.testing.assert_allclose(r0["params"][keep], one["params"][keep], rtol=2e-4, atol=2e-6)   # ... and equal the one-process run
    np.testing.assert_allclose(r0["totals"], one["totals"], rtol=1e-9)              # global exchange steps / hits / sample-steps
    np.testing.assert_array_equal(r0["totals"], r1["totals"])
    # rank 0 alone logs and checkpoints; the log carries the GLOBAL minibatch's figures (same lines as the one-process log)
    logs = os.path.join(tmp, "logs_dp")
    assert os.path.exists(os.path.join(logs, "dp.log")) and os.path.exists(os.path.join(logs, "dp.pt"))
    grab = lambda path: [ln.split("] ", 1)[1] for ln in open(path) if "Training Accuracy" in ln or "Development Accuracy" in ln]
    a, b = grab(os.path.join(logs, "dp.log")), grab(os.path.join(tmp, "logs_one", "one.log"))
    assert len(a) == len(b) >= 3
    for la, lb in zip(a, b):
        assert la.rsplit(": ", 1)[0] == lb.rsplit(": ", 1)[0]
        assert abs(float(la.rsplit(": ", 1)[1]) - float(lb.rsplit(": ", 1)[1])) < 1e-6, (la, lb)


def test_sharded_loader_partitions_every_global_batch(tmp_path):
    """load_hdf5(shard=(rank, world)): the union of the ranks' rows is the single-process batch, in its (sorted) order."""
    from multimodalgame_amd import misc
    paths = misc.write_synthetic_dataset(str(tmp_path), n_classes=N_CLASSES, per_class=PER_CLASS, feat_dim=16, wv_dim=7)
    for epoch in (0, 3):
        full = list(misc.load_hdf5(paths["train_file"], BATCH, epoch, True, device="cpu", with_ids=False))
        parts = [list(misc.load_hdf5(paths["train_file"], BATCH, epoch, True, device="cpu", with_ids=False, shard=(r, 3))) for r in range(3)]
        assert len(full) == len(parts[0]) == 4
        for i, b in enumerate(full):
            np.testing.assert_array_equal(torch.cat([p[i]["avgpool_512"] for p in parts]).numpy(), b["avgpool_512"].numpy())
            np.testing.assert_array_equal(torch.cat([p[i]["target"] for p in parts]).numpy(), b["target"].numpy())
    with pytest.raises(AssertionError):
        list(misc.load_hdf5(paths["train_file"], 10, 0, True, device="cpu", shard=(0, 3)))      # 10 samples over 3 ranks
