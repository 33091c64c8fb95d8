"""End-to-end run of the reference's command line on the GPU path: synthetic HDF5 / descriptions / GloVe
files in the reference's formats, a few optimizer steps, log lines, checkpoint save -> resume -> -eval_only."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


def _argv(tmp, name, extra=()):
    return ["model.py", "-experiment_name", name, "-model_type", "Adaptive", "-max_exchange", "10", "-batch_size", "64",
            "-rec_w_dim", "32", "-sender_out_dim", "32", "-img_h_dim", "256", "-rec_hidden", "64", "-learning_rate", "1e-4",
            "-entropy_rec", "0.01", "-entropy_sen", "0.01", "-entropy_s", "0.08", "-use_binary", "-max_epoch", "1",
            "-top_k_dev", "6", "-top_k_train", "6", "-wv_dim", "100", "-log_path", os.path.join(tmp, "logs"),
            "-synthetic_data", os.path.join(tmp, "data"), "-log_interval", "5", "-log_dev", "10", "-save_after", "10",
            "-save_interval", "10", "-exchange_samples", "2"] + list(extra)


def test_train_checkpoint_resume_eval(tmp_path):
    from multimodalgame_amd import model, flags
    tmp = str(tmp_path)
    flags.define_flags(); flags.FLAGS.Reset()
    model.main(_argv(tmp, "demo", ["-max_steps", "21"]))
    log = open(os.path.join(tmp, "logs", "demo.log")).read()
    for pat in (r"\[1\] Starting epoch: 0", r"Epoch: 0 Step: 20 Batch: 20 Training Accuracy: ", r"Loss Receiver \(S\): ",
                r"Loss Baseline \(R\): ", r"Development Accuracy: ", r"Conversation Length \(avg/std\): ", r"Checkpointing\.",
                r"Train:\n"):
        assert re.search(pat, log), pat
    ck = torch.load(os.path.join(tmp, "logs", "demo.pt"), weights_only=False)
    assert set(ck.keys()) == {"data", "optimizers", "models"}                     # misc.py:65-69
    assert set(ck["models"].keys()) == {"receiver", "sender", "baseline_rec", "baseline_sen"}
    assert set(ck["optimizers"].keys()) == {"optimizer_rec", "optimizer_sen", "optimizer_bas_rec", "optimizer_bas_sen"}
    assert ck["data"]["step"] == 20 and "rnn.weight_ih" in ck["models"]["receiver"]
    assert "square_avg" in ck["optimizers"]["optimizer_rec"]["state"][0]
    assert os.path.exists(os.path.join(tmp, "logs", "demo.json")) and os.path.exists(os.path.join(tmp, "logs", "demo.conf_mat.txt"))
    # resume + eval_only through -log_load (README.md:57-68 workflow)
    flags.FLAGS.Reset()
    model.main(["model.py", "-log_load", os.path.join(tmp, "logs", "demo.json"), "-eval_only", "-experiment_name", "demo-eval",
                "-checkpoint", os.path.join(tmp, "logs", "demo.pt"), "-log_path", os.path.join(tmp, "logs")])
    # -log_load restores every flag of the training run, derived file names included (model.py:1745-1750)
    csv = open(os.path.join(tmp, "logs", "demo.eval.csv")).read().splitlines()
    assert csv[0] == "checkpoint,eval_file,topk,step,best_dev_acc,eval_acc,convlen_mean,convlen_std"
    assert csv[1].split(",")[3] == "20"
    flags.FLAGS.Reset()


def test_binary_only_message_dump(tmp_path):
    """README.md:57-68 workflow: train briefly, then -binary_only on a dev file whose batches hold one class each."""
    import numpy as np
    from multimodalgame_amd import model, flags, hdf5io
    from multimodalgame_amd.binary_vectors import record_types
    tmp = str(tmp_path)
    flags.define_flags(); flags.FLAGS.Reset()
    model.main(_argv(tmp, "bv", ["-max_steps", "11", "-max_exchange", "4"]))
    rs = np.random.RandomState(3)
    dev = os.path.join(tmp, "dev_sorted.hdf5")
    with hdf5io.File(dev, "w") as f:                       # 3 classes x 10 samples, sorted: every batch of 10 has one target
        f.write("avgpool_512", np.abs(rs.standard_normal((30, 1, 512))).astype(np.float32))
        f.write("Target", np.repeat(np.arange(3), 10).astype(np.int32))
        f.write("Location", np.array([("d%02d.jpg" % i).encode() for i in range(30)], dtype="S50"))
    flags.FLAGS.Reset()
    out = os.path.join(tmp, "logs", "bv.bv.hdf5")
    model.main(["model.py", "-log_load", os.path.join(tmp, "logs", "bv.json"), "-binary_only", "-checkpoint",
                os.path.join(tmp, "logs", "bv.pt"), "-dev_file", dev, "-batch_size_dev", "10", "-fixed_exchange",
                "-binary_output", out, "-log_path", os.path.join(tmp, "logs")])
    comm_t, preds_t = record_types(32, 30)
    with hdf5io.File(out, "r") as f:
        comm, preds = f.read_struct("Communication", comm_t), f.read_struct("Predictions", preds_t)
    T = 4
    assert len(comm) == 30 * T * 2 and len(preds) == 30 * T                 # one record per agent message / receiver step
    assert set(comm["AgentId"].tolist()) == {b"S", b"R"} and set(comm["Index"].tolist()) == set(range(2 * T))
    assert set(np.unique(comm["BinaryVec"]).tolist()) <= {0.0, 1.0}
    assert ((comm["BinaryProb"] >= 0) & (comm["BinaryProb"] <= 1)).all()
    np.testing.assert_array_equal(np.round(comm["BinaryProb"]), comm["BinaryVec"])   # eval mode: round(p)  (model.py:229, 462)
    assert comm["ExampleId"][0] == b"d00.jpg" and (preds["Target"][:10 * 1] == 0).all()
    flags.FLAGS.Reset()


@pytest.mark.parametrize("kind", ["RMSprop", "Adam"])
def test_optimizer_state_roundtrip_with_torch_optim(kind):
    """'optimizer_*' entries of a checkpoint are torch.optim state_dicts numbered like module.parameters() (Sender: code_bias
    first): a state built by a real torch.optim object on the module loads into the flat engine state by NAME, and what
    FlatOptimizer saves loads back into torch.optim."""
    from multimodalgame_amd.agents import Baseline, Receiver, Sender
    from multimodalgame_amd.game import Game

    class Fl(object):
        max_exchange, fixed_exchange, s_prob_prod = 3, False, True
        entropy_s = entropy_sen = entropy_rec = None
        first_rec, optim_type, learning_rate, top_k_train = 0.0, kind, 1e-3, 2
    sender = Sender("avgpool_512", 16, 8, 4, 4, True)
    receiver = Receiver(4, 8, 4, 1, 4, 1, True)
    game = Game(sender, receiver, Baseline(12, 8, 4, 0), Baseline(12, 0, 4, 4), flags=Fl(), device="cuda:0")
    eng = game.engine_for(8, 5)
    names = [n for n, _ in sender.named_parameters()]
    assert names[0] == "code_bias"
    # a reference-style optimizer on CPU copies of the sender's parameters, one step with known gradients
    ref_params = [torch.nn.Parameter(p.detach().cpu().clone()) for _, p in sender.named_parameters()]
    opt = getattr(torch.optim, kind)(ref_params, lr=1e-3)
    for i, p in enumerate(ref_params):
        p.grad = torch.full_like(p, 0.01 * (i + 1))
    opt.step()
    fo = game.optimizers_dict()["optimizer_sen"]
    fo.load_state_dict(opt.state_dict())
    key = "square_avg" if kind == "RMSprop" else "exp_avg"
    for i, n in enumerate(names):
        view = eng.params["sender"][n]
        off = view.storage_offset()
        got = eng.opt_state[off:off + view.numel()].view(view.shape).cpu()
        torch.testing.assert_close(got, opt.state_dict()["state"][i][key])
    sd = fo.state_dict()
    opt2 = getattr(torch.optim, kind)([torch.nn.Parameter(p.detach().cpu().clone()) for p in ref_params], lr=1e-3)
    opt2.load_state_dict(sd)
    for i in range(len(names)):
        torch.testing.assert_close(opt2.state_dict()["state"][i][key], opt.state_dict()["state"][i][key])


def _dp_worker(rank, world, port, argv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    from multimodalgame_amd import model, flags
    flags.define_flags(); flags.FLAGS.Reset()
    model.main(argv)


@pytest.mark.parametrize("mode", ["adaptive", "continuous"])
def test_data_parallel_model_run(mode, tmp_path):
    """the OUTER EPOCH LOOP sharded over the ranks.  Two gloo ranks of `python -m multimodalgame_amd.model`
    share this box's GPU (LOCAL_RANK % device_count), each trains on its 32 rows of every 64-sample minibatch of the
    reference's batch order; rank 0 alone logs / evaluates / checkpoints.  The checkpoint after 12 optimizer steps must hold
    the parameters of the one-process run (in-kernel Philox sampling is keyed by the global sample index), and the log lines
    the losses / training accuracy of the global minibatch.  (The CPU twin of this test, with an oracle-backed engine, is
    tests/test_dp_epoch_loop.py.)"""
    import socket
    import torch.multiprocessing as mp
    from multimodalgame_amd import model, flags
    tmp = str(tmp_path)
    extra = ["-max_steps", "12", "-save_after", "0", "-save_interval", "1", "-exchange_samples", "0"]
    if mode == "continuous":
        extra += ["-nouse_binary", "-model_type", "Fixed", "-max_exchange", "4"]
    flags.define_flags(); flags.FLAGS.Reset()
    model.main(_argv(tmp, "one", extra))
    flags.FLAGS.Reset()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_dp_worker, args=(2, port, _argv(tmp, "dp", extra + ["-dist_backend", "gloo"])), nprocs=2, join=True)
    one = torch.load(os.path.join(tmp, "logs", "one.pt"), weights_only=False)
    dp = torch.load(os.path.join(tmp, "logs", "dp.pt"), weights_only=False)
    assert one["data"]["step"] == dp["data"]["step"] == 11
    for agent, sd in one["models"].items():
        for k, v in sd.items():
            if agent == "receiver" and k == "y2.bias":
                continue                                  # exact gradient 0: see tests/common.py
            torch.testing.assert_close(dp["models"][agent][k], v, rtol=2e-4, atol=2e-6, msg="%s.%s" % (agent, k))
    grab = lambda name: [ln.split("] ", 1)[1] for ln in open(os.path.join(tmp, "logs", name + ".log"))
                         if "Training Accuracy" in ln or "Loss Receiver (Y)" in ln]
    a, b = grab("dp"), grab("one")
    assert len(a) == len(b) >= 4 and "Data parallel: 2 ranks, 32 samples of every 64-sample minibatch per rank" in open(os.path.join(tmp, "logs", "dp.log")).read()
    for la, lb in zip(a, b):
        assert la.rsplit(": ", 1)[0] == lb.rsplit(": ", 1)[0]
        assert abs(float(la.rsplit(": ", 1)[1]) - float(lb.rsplit(": ", 1)[1])) < 1e-4, (la, lb)


def test_deferred_log_block_equals_the_synchronous_one(tmp_path, monkeypatch):
    """model.run() enqueues a log minibatch's block (device-side reductions, sample-dump slices, one non-blocking copy to pinned
    memory) and writes it when the copy has landed, a few minibatches later; MMG_LOG_SYNC=1 copies and writes at once, as rounds
    1-4 did.  Same seed, same data: the two log files must be identical line by line -- loss lines, Predictions, the three
    Entropy blocks, Train / Eval sample dumps, dev evaluation and checkpoint lines in between, in the same order."""
    from multimodalgame_amd import model, flags
    logs = {}
    for name, sync in (("deferred", None), ("sync", "1")):
        tmp = str(tmp_path / name)
        if sync:
            monkeypatch.setenv("MMG_LOG_SYNC", sync)
        flags.define_flags(); flags.FLAGS.Reset()
        model.main(_argv(tmp, "run", ["-max_steps", "23", "-log_interval", "3"]))
        flags.FLAGS.Reset()
        text = open(os.path.join(tmp, "logs", "run.log")).read()
        logs[name] = [re.sub(r"^\d\d-\d\d-\d\d \d\d:\d\d:\d\d ", "", l) for l in text.replace(tmp, "<tmp>").splitlines() if "Flag Values" not in l]   # (FileLogger's time stamp)
    a, b = logs["deferred"], logs["sync"]
    assert len(a) == len(b) and len(a) > 200
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y and "<tmp>" not in x]
    assert not diff, diff[:5]
    assert sum(1 for l in a if "Training Accuracy" in l) == 8 and any("Eval:" in l for l in a)


def test_runs_of_minibatches_in_one_library_call_write_the_same_log(tmp_path, monkeypatch):
    """Round 6: model.run() hands every run of plain minibatches between two log / evaluation / checkpoint steps to ONE library call
    (mmg_train_steps over the epoch's batch-ordered gather, misc.Epoch); MMG_LOOP_PER_STEP=1 keeps the per-minibatch loop of rounds
    1-5 (a Python generator step + one ctypes call per minibatch).  Same seed, same data: identical log files line by line and
    identical checkpoints, across an epoch boundary (46 minibatches per epoch) and a resume."""
    from multimodalgame_amd import model, flags
    logs, cks = {}, {}
    for name, per_step in (("runs", None), ("per_step", "1")):
        tmp = str(tmp_path / name)
        if per_step:
            monkeypatch.setenv("MMG_LOOP_PER_STEP", per_step)
        flags.define_flags(); flags.FLAGS.Reset()
        extra = ["-max_epoch", "3", "-log_interval", "7", "-log_dev", "40", "-save_after", "20", "-save_interval", "20"]
        model.main(_argv(tmp, "run", extra + ["-max_steps", "61"]))
        flags.FLAGS.Reset()
        model.main(_argv(tmp, "run", extra + ["-max_steps", "75"]))           # resumes from step 60's checkpoint
        flags.FLAGS.Reset()
        text = open(os.path.join(tmp, "logs", "run.log")).read()
        logs[name] = [re.sub(r"^\d\d-\d\d-\d\d \d\d:\d\d:\d\d ", "", l) for l in text.replace(tmp, "<tmp>").splitlines() if "Flag Values" not in l]
        cks[name] = torch.load(os.path.join(tmp, "logs", "run.pt"), weights_only=False)
    monkeypatch.delenv("MMG_LOOP_PER_STEP")
    a, b = logs["runs"], logs["per_step"]
    assert len(a) == len(b) and len(a) > 200
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y and "<tmp>" not in x]
    assert not diff, diff[:5]
    assert any("Starting epoch: 1" in l for l in a) and any("Loaded at step: 60" in l for l in a)
    assert cks["runs"]["data"] == cks["per_step"]["data"]
    for agent, sd in cks["runs"]["models"].items():
        for k, v in sd.items():
            assert torch.equal(v, cks["per_step"]["models"][agent][k]), (agent, k)
