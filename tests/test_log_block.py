"""The log block of model.run() (reference model.py:1342-1518) on the CPU: the loss lines, "Predictions", the three "Entropy ..."
blocks and the Train / Eval sample dumps in the reference's order and layout, written from the tape of a run-all log minibatch.
The engine under the loop is the oracle-backed stand-in of tests/oracle_engine.py (swapped in HERE; the product has no CPU path)."""
import math
import os
import re

import numpy as np
import torch

from tests import test_dp_epoch_loop as T


def test_log_block_lines_and_order(tmp_path, monkeypatch):
    from multimodalgame_amd import flags as _flags, game as _game, misc, model as _model
    from tests import oracle_engine
    monkeypatch.setattr(_game, "Engine", oracle_engine.OracleEngine)
    monkeypatch.setattr(_model, "_device", lambda local_rank: torch.device("cpu"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    tmp = str(tmp_path)
    paths = misc.write_synthetic_dataset(os.path.join(tmp, "data"), n_classes=T.N_CLASSES, per_class=T.PER_CLASS, feat_dim=16, wv_dim=7)
    monkeypatch.setattr(T, "PATHS", dict(paths))
    argv = T._argv(tmp, "logblock", T.MODES["adaptive"], extra=["-max_epoch", "1"])
    argv[argv.index("-exchange_samples") + 1] = "2"
    oracle_engine.OracleEngine.instances = []
    _flags.define_flags(); _flags.FLAGS.Reset(); _flags.FLAGS(argv)
    _flags.FLAGS.img_feat_dim = 16
    _flags.default_flags(argv)
    _flags.FLAGS.img_feat_dim = 16
    captured = []
    real = _model._entropy_lines

    def spy(eng, target, L, snap=None):
        out = real(eng, target, L, snap)
        # independent recomputation of step 0 of the sender's entropy: every sample is active at step 0 (model.py:919-923)
        p = eng.tape["pz"][0].double()
        want0 = -float((p * torch.log(p + 1e-8) + (1 - p) * torch.log(1 - p + 1e-8)).sum(1).mean())
        y0 = torch.softmax(eng.tape["y"][0].double(), 1)
        captured.append((want0, -float((torch.log(y0 + 1e-8) * y0).sum(1).mean()), int(L["n_steps"])))
        return out
    monkeypatch.setattr(_model, "_entropy_lines", spy)
    try:
        _model.run()
    finally:
        _flags.FLAGS.Reset()
    text = open(os.path.join(tmp, "logs_logblock", "logblock.log")).read()
    blocks = text.split("Training Accuracy")[1:]
    assert len(blocks) == 2 and len(captured) == 2            # steps 0 and 3 of the 4 minibatches (-log_interval 3)
    for blk, (want_sen0, want_y0, n) in zip(blocks, captured):
        order = [blk.index(k) for k in ("Loss Sender:", "Loss Receiver (Y):", "Loss Receiver (Z):", "Loss Receiver (S):", "Loss Baseline (S):",
                                        "Loss Baseline (R):", "Predictions: ", "Entropy Sender Binary", "Entropy Receiver Binary",
                                        "Entropy Receiver Predictions", "Train:", "Eval:")]
        assert order == sorted(order), "log block out of the reference's order (model.py:1348-1518)"
        sen = re.search(r"Entropy Sender Binary((?:\n\d+\. [-0-9.e]+)+)\n", blk).group(1).strip().split("\n")
        rec = re.search(r"Entropy Receiver Binary((?:\n\d+\. [-0-9.e]+)+)\n", blk).group(1).strip().split("\n")
        yy = re.search(r"Entropy Receiver Predictions((?:\n\d+\. [-0-9.e]+)+)\n", blk).group(1).strip().split("\n")
        assert len(sen) == n and len(rec) == n - 1 and len(yy) == n         # rec_feats[:-1] (model.py:1284-1289)
        assert [int(l.split(". ")[0]) for l in sen] == list(range(n))
        assert abs(float(sen[0].split(". ")[1]) - want_sen0) < 1e-5
        assert abs(float(yy[0].split(". ")[1]) - want_y0) < 1e-5
        for l in sen + rec:
            assert 0.0 <= float(l.split(". ")[1]) <= 6 * math.log(2) + 1e-6     # W = 6 bits: entropy within [0, W ln 2]
        # Predictions: a [2, B] integer tensor, targets over predictions
        pred = re.search(r"Predictions: tensor\(\[\[([^\]]*)\],\s*\[([^\]]*)\]\]\)", blk)
        assert pred and len(pred.group(1).split(",")) == T.BATCH == len(pred.group(2).split(","))
        # sample dumps: `n` steps for each of the two samples, a bit string of W = 6 per agent and the s= flag (last one forced 0)
        tr = blk[blk.index("Train:"):blk.index("Eval:")]
        rows = re.findall(r"\n\s+(\d+) S: ([01]{6})\s+[0-9.]+\s+s=([01]) R: ([01]{6})", tr)
        assert len(rows) == 2 * n and [int(r[0]) for r in rows] == list(range(n)) * 2
        assert rows[n - 1][2] == "0" and rows[-1][2] == "0"
        ev = blk[blk.index("Eval:"):]
        ev_rows = re.findall(r"\n\s+(\d+) S: ([01]{6})\s+[0-9.]+\s+s=([01]) R: ([01]{6})", ev)
        assert len(ev_rows) >= 2 and ev_rows[-1][2] == "0"
