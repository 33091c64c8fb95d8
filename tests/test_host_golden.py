"""Rows f3 / f4 of SURVEY.md §8 against golden vectors produced by the reference's own functions
(tests/golden/make_golden_host.py)."""
import os

import numpy as np
import pytest
import torch

from multimodalgame_amd import misc
from tests import common

FIX = os.path.join(common.GOLDEN_DIR, "fixtures")


def test_description_pipeline_matches_reference():
    """read_data -> embed -> cbow on the committed mini CSV / GloVe snippet = the reference's misc.py:229-254, 306-340
    (same tokenizer and stop-word list on both sides; the reference orders words by set() hash, so sets are compared)."""
    z = np.load(os.path.join(common.GOLDEN_DIR, "g6_desc_pipeline.npz"))
    descr, word_dict, dict_size, label_id_to_idx, idx_to_label = misc.read_data(os.path.join(FIX, "descriptions_mini.csv"))
    word_dict = misc.embed(word_dict, os.path.join(FIX, "glove_mini.8d.txt"))
    descr = misc.cbow(descr, word_dict)
    n = int(z["n_classes"])
    assert len(descr) == n and dict_size == int(z["dict_size"])
    assert [descr[i]["name"] for i in range(n)] == list(z["names"])
    assert ["|".join(sorted(descr[i]["desc"])) for i in range(n)] == list(z["desc_sorted"])
    assert sorted(word_dict) == list(z["vocab_sorted"])
    assert [int(word_dict[w]["emb"] is not None) for w in sorted(word_dict)] == list(z["vocab_found"])
    assert sorted(word_dict[w]["id"] for w in word_dict) == list(range(1, dict_size + 1))
    assert [label_id_to_idx[int(k)] for k in z["label_ids"]] == list(z["label_rows"])
    assert [idx_to_label[i] for i in range(n)] == list(z["idx_to_label"])
    np.testing.assert_allclose(torch.stack([descr[i]["cbow"] for i in range(n)]).numpy(), z["cbow"], atol=1e-6)
    np.testing.assert_allclose(torch.stack([descr[i]["set"].sum(0) for i in range(n)]).numpy(), z["set_rowsum"], atol=1e-5)
    for i in range(n):
        assert descr[i]["set"].shape == (len(descr[i]["desc"]), 8)


def test_checkpoint_roundtrip_and_build_mask(tmp_path):
    lin = torch.nn.Linear(3, 2)
    opt = torch.optim.RMSprop(lin.parameters(), lr=0.1)
    lin(torch.ones(1, 3)).sum().backward(); opt.step()
    path = str(tmp_path / "c.pt")
    misc.torch_save(path, dict(step=7, best_dev_acc=0.5), {"m": lin}, {"o": opt})
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"data", "models", "optimizers"} and ck["data"]["step"] == 7           # misc.py:58-69 layout
    assert all(not t.is_cuda for t in ck["models"]["m"].values())
    lin2 = torch.nn.Linear(3, 2); opt2 = torch.optim.RMSprop(lin2.parameters(), lr=0.1)
    data = misc.torch_load(path, {"m": lin2}, {"o": opt2})
    assert data["best_dev_acc"] == 0.5
    torch.testing.assert_close(lin2.weight, lin.weight)
    torch.testing.assert_close(opt2.state_dict()["state"][0]["square_avg"], opt.state_dict()["state"][0]["square_avg"])
    with pytest.raises(Exception):
        misc.torch_load(str(tmp_path / "missing.pt"), {}, {})
    assert misc.build_mask("0:2,4,6:8", 9).view(-1).tolist() == [1, 1, 0, 0, 1, 0, 1, 1, 0]


@pytest.mark.gpu
def test_binary_vectors_match_reference(tmp_path):
    """-binary_only records vs the reference's extract_binary (binary_vectors.py:12-135) on the same weights and batch:
    Index convention (sender 2i, receiver 2i + 1), Rank (:98), bits exact, probabilities / scores at 1e-5."""
    from multimodalgame_amd import binary_vectors as bv, hdf5io
    from multimodalgame_amd.agents import Baseline, Receiver, Sender
    from multimodalgame_amd.game import Game
    from oracle import cpu_ref
    z, meta = common.load_golden("g7_binary_vectors")
    fl = common.flags_from_meta(meta)

    class Fl(object):
        pass
    F = Fl()
    for k, v in fl.__dict__.items():
        setattr(F, k, v)
    F.img_feat, F.binary_output = "avgpool_512", str(tmp_path / "bv.hdf5")
    F.desc_attn, F.sender_mix, F.ignore_receiver, F.ignore_code, F.visual_attn, F.bit_flip = False, "sum", False, False, False, False
    sender = Sender("avgpool_512", fl.img_feat_dim, fl.img_h_dim, fl.rec_w_dim, fl.sender_out_dim, True)
    receiver = Receiver(fl.sender_out_dim, fl.wv_dim, fl.rec_hidden, 1, fl.rec_w_dim, 1, True)
    bas_s = Baseline(fl.baseline_hid_dim, fl.img_h_dim, fl.rec_w_dim, 0)
    bas_r = Baseline(fl.baseline_hid_dim, 0, fl.rec_w_dim, fl.rec_hidden)
    game = Game(sender, receiver, bas_s, bas_r, flags=F, device="cuda:0")
    B, D = int(z["batch"]), int(z["n_classes"])
    eng = game.engine_for(B, D)
    shapes = {a: {k: tuple(v.shape) for k, v in d.items()} for a, d in eng.params.items()}
    eng.load_state_dicts(cpu_ref.fill_state_dicts(shapes, seed=int(z["seed_weights"])))
    eng.params["receiver"]["s.bias"].fill_(1.2)
    x, _, desc = cpu_ref.synthetic_batch(B, D, fl.img_feat_dim, fl.wv_dim, seed=int(z["seed_data"]))
    dev = str(tmp_path / "dev.hdf5")
    with hdf5io.File(dev, "w") as f:
        f.write("avgpool_512", x.reshape(B, 1, -1).astype(np.float32))
        f.write("Target", z["target"].astype(np.int32))
        f.write("Location", z["example_ids"].astype("S50"))
    n_comm, n_pred = bv.extract_binary(F, dev, B, 0, False, sender, receiver, torch.from_numpy(desc).cuda(), int, torch.device("cuda:0"))
    comm_t, preds_t = bv.record_types(fl.sender_out_dim, D)
    with hdf5io.File(F.binary_output, "r") as f:
        comm, preds = f.read_struct("Communication", comm_t), f.read_struct("Predictions", preds_t)
    assert len(comm) == len(z["comm_index"]) and len(preds) == len(z["pred_index"])
    np.testing.assert_array_equal(comm["Index"], z["comm_index"])
    np.testing.assert_array_equal(comm["AgentId"], z["comm_agent"])
    np.testing.assert_array_equal(comm["Target"], z["comm_target"])
    np.testing.assert_array_equal(comm["Rank"], z["comm_rank"])
    np.testing.assert_array_equal(comm["BinaryVec"], z["comm_vec"])
    np.testing.assert_allclose(comm["BinaryProb"], z["comm_prob"], atol=1e-5)
    np.testing.assert_array_equal(np.char.strip(comm["ExampleId"]), z["comm_ids"])
    np.testing.assert_array_equal(preds["Index"], z["pred_index"])
    np.testing.assert_array_equal(preds["Rank"], z["pred_rank"])
    np.testing.assert_allclose(preds["Predictions"], z["pred_scores"], atol=1e-5)
    np.testing.assert_allclose(preds["StopProb"], z["pred_stop_prob"], atol=1e-5)
    np.testing.assert_array_equal(preds["StopVec"], z["pred_stop_vec"])
    np.testing.assert_array_equal(preds["StopMask"], z["pred_stop_mask"])


def _eval_dev_vs_reference(tmp_path, monkeypatch, dev):
    """model.eval_dev against the reference's own eval_dev (model.py:580-722) on the same weights and the same two dev batches
    (the second one short): accuracy with the NOMINAL batch size in the denominator (:667), conversation-length mean / std,
    mean Hamming distances of both agents' messages, and the confusion matrix as sklearn lays it out (only classes that occur)."""
    from multimodalgame_amd import flags as _flags, model
    from multimodalgame_amd.agents import Baseline, Receiver, Sender
    from multimodalgame_amd.game import Game
    from oracle import cpu_ref
    z, meta = common.load_golden("g8_eval_dev")
    fl = common.flags_from_meta(meta)
    _flags.define_flags(); _flags.FLAGS.Reset()
    argv = ["model.py", "-model_type", "Adaptive", "-max_exchange", str(fl.max_exchange), "-rec_w_dim", str(fl.rec_w_dim),
            "-sender_out_dim", str(fl.sender_out_dim), "-img_h_dim", str(fl.img_h_dim), "-rec_hidden", str(fl.rec_hidden),
            "-wv_dim", str(fl.wv_dim), "-baseline_hid_dim", str(fl.baseline_hid_dim), "-use_binary", "-top_k_dev", "2",
            "-log_path", str(tmp_path)]
    _flags.FLAGS(argv)
    _flags.default_flags(argv)
    F = _flags.FLAGS
    F.img_feat_dim = fl.img_feat_dim
    sender = Sender("avgpool_512", fl.img_feat_dim, fl.img_h_dim, fl.rec_w_dim, fl.sender_out_dim, True, False, 0, False, 0)
    receiver = Receiver(fl.sender_out_dim, fl.wv_dim, fl.rec_hidden, 1, fl.rec_w_dim, 1, True)
    game = Game(sender, receiver, Baseline(fl.baseline_hid_dim, fl.img_h_dim, fl.rec_w_dim, 0),
                Baseline(fl.baseline_hid_dim, 0, fl.rec_w_dim, fl.rec_hidden), device=dev)
    B, D = int(z["batch"]), int(z["n_classes"])
    eng = game.engine_for(B, D)
    shapes = {a: {k: tuple(v.shape) for k, v in d.items()} for a, d in eng.params.items()}
    for a, d in cpu_ref.fill_state_dicts(shapes, seed=int(z["seed_weights"])).items():
        for k, v in d.items():
            eng.params[a][k].copy_(torch.as_tensor(v, dtype=torch.float32).view_as(eng.params[a][k]))
    eng.params["receiver"]["s.bias"].fill_(float(z["s_bias"]))
    sizes = [int(v) for v in z["sizes"]]
    x0, _, desc = cpu_ref.synthetic_batch(sizes[0], D, fl.img_feat_dim, fl.wv_dim, seed=int(z["seed_data"]))
    x1, _, _ = cpu_ref.synthetic_batch(sizes[1], D, fl.img_feat_dim, fl.wv_dim, seed=int(z["seed_data"]) + 1)
    dev = torch.device(dev)

    def fake_load_hdf5(dev_file, batch_size, epoch, shuffle, truncate_final_batch=False, map_labels=int, feats=(), device=None, **kw):
        assert truncate_final_batch and batch_size == B
        for x, t in ((x0, z["target0"]), (x1, z["target1"])):
            yield {"target": torch.from_numpy(t.astype(np.int64)).to(dev), "avgpool_512": torch.from_numpy(x).to(dev)}
    monkeypatch.setattr(model, "load_hdf5", fake_load_hdf5)
    conf_path = str(tmp_path / "conf.txt")
    acc, extra = model.eval_dev("dev", B, 0, False, 2, game, torch.from_numpy(desc).to(dev), int, conf_path, dev)
    assert acc == pytest.approx(float(z["accuracy"]), abs=1e-12)
    for k in ("conversation_lengths_mean", "conversation_lengths_std"):
        assert float(extra[k]) == pytest.approx(float(z[k]), abs=1e-9), k
    for k in ("hamming_sen_mean", "hamming_rec_mean"):
        assert float(extra[k]) == pytest.approx(float(z[k]), abs=1e-6), k
    np.testing.assert_array_equal(np.loadtxt(conf_path, delimiter=",", dtype=np.int64, ndmin=2), z["conf_mat"])
    _flags.FLAGS.Reset()


@pytest.mark.gpu
def test_eval_dev_matches_reference(tmp_path, monkeypatch):
    _eval_dev_vs_reference(tmp_path, monkeypatch, "cuda:0")


def test_eval_dev_host_logic_matches_reference_cpu(tmp_path, monkeypatch):
    """The same comparison without a GPU: the host side of eval_dev (nominal-batch denominator, conversation lengths, Hamming
    means, sklearn-style confusion matrix) over the oracle-backed engine stand-in (tests/oracle_engine.py, swapped in here)."""
    from multimodalgame_amd import game
    from tests import oracle_engine
    monkeypatch.setattr(game, "Engine", oracle_engine.OracleEngine)
    _eval_dev_vs_reference(tmp_path, monkeypatch, "cpu")
