"""Host-side pieces around the path ("next" rows of SURVEY.md §8f): HDF5 schema round trip through the
ctypes->libhdf5 binding, the reference's batch-order semantics, description pipeline, logger format."""
import random
import re

import numpy as np
import pytest
import torch

from multimodalgame_amd import hdf5io, misc


def test_hdf5_roundtrip_reference_schema(tmp_path):
    p = str(tmp_path / "f.hdf5")
    feats = np.abs(np.random.RandomState(0).standard_normal((10, 1, 8))).astype(np.float32)
    tgt = np.arange(10, dtype=np.int32)[::-1].copy()
    loc = np.array([("img%d.jpg" % i).encode() for i in range(10)], dtype="S50")
    with hdf5io.File(p, "w") as f:
        f.write("avgpool_512", feats); f.write("Target", tgt); f.write("Location", loc)
    with hdf5io.File(p, "r") as f:
        assert "Target" in f and "nope" not in f
        assert f.shape("avgpool_512") == (10, 1, 8)
        np.testing.assert_array_equal(f.read("avgpool_512"), feats)
        np.testing.assert_array_equal(f.read("Target"), tgt)
        np.testing.assert_array_equal(f.read("Location"), loc)
    with pytest.raises(hdf5io.Hdf5Error):
        hdf5io.File(str(tmp_path / "missing.hdf5"), "r")


def test_load_hdf5_order_semantics(tmp_path):
    paths = misc.write_synthetic_dataset(str(tmp_path), n_classes=3, per_class=7, feat_dim=4, wv_dim=5)
    batches = list(misc.load_hdf5(paths["train_file"], 4, 2, True))
    assert len(batches) == 21 // 4                                           # drop-last (misc.py:274)
    order = list(range(21)); random.seed(11 + 2); random.shuffle(order)      # misc.py:270-271
    with hdf5io.File(paths["train_file"]) as f:
        tgt, x = f.read("Target"), f.read("avgpool_512")
    for i, b in enumerate(batches):
        idx = sorted(order[i * 4:(i + 1) * 4])                               # misc.py:282
        np.testing.assert_array_equal(b["target"].numpy(), tgt[idx])
        np.testing.assert_array_equal(b["avgpool_512"].numpy(), x[idx, 0])   # squeezed (misc.py:296)
    assert len(list(misc.load_hdf5(paths["dev_file"], 4, 0, False, truncate_final_batch=True))) == 6
    assert list(misc.load_hdf5(paths["dev_file"], 4, 0, False, truncate_final_batch=True))[-1]["target"].shape[0] == 1


def test_load_hdf5_device_resident_equals_host_path(tmp_path):
    """device=...: the file is preloaded once, one gather per epoch lays the samples out in batch order, a batch is two
    views -- same batches, same order, same mapped targets as the per-batch host path (misc.py:257-302 semantics)."""
    paths = misc.write_synthetic_dataset(str(tmp_path), n_classes=5, per_class=9, feat_dim=6, wv_dim=5)
    label_map = {c: (c * 3) % 5 for c in range(5)}
    calls = []
    fn = lambda v: calls.append(v) or label_map.get(v)
    for epoch, shuffle, trunc in ((0, True, False), (3, True, False), (1, False, True)):
        host = list(misc.load_hdf5(paths["train_file"], 8, epoch, shuffle, truncate_final_batch=trunc, map_labels=label_map.get))
        calls.clear()
        dev = list(misc.load_hdf5(paths["train_file"], 8, epoch, shuffle, truncate_final_batch=trunc, map_labels=fn, device="cpu"))
        assert len(host) == len(dev) == (45 // 8 + (1 if trunc else 0))
        assert len(calls) <= 5                                               # map_labels runs per distinct label, once per file
        for a, b in zip(host, dev):
            assert torch.equal(a["target"], b["target"]) and b["target"].dtype == torch.int64
            assert torch.equal(a["avgpool_512"], b["avgpool_512"]) and b["avgpool_512"].dim() == 2
            np.testing.assert_array_equal(a["example_ids"], b["example_ids"])
    # views of ONE per-epoch gather: consecutive batches share storage; with_ids=False drops the string array
    dev = list(misc.load_hdf5(paths["train_file"], 8, 0, True, map_labels=fn, device="cpu", with_ids=False))
    assert "example_ids" not in dev[0]
    assert dev[1]["avgpool_512"].data_ptr() == dev[0]["avgpool_512"].data_ptr() + 8 * 6 * 4


def test_description_pipeline(tmp_path):
    csv = tmp_path / "d.csv"
    csv.write_text("7,agama,small terrestrial lizard of warm regions, of the Old World\n3,drake,adult male of a wild duck\n")
    glove = tmp_path / "g.txt"
    glove.write_text("small 1 0\nlizard 0 1\nduck 2 2\nmale 0 4\n")
    descr, word_dict, dict_size, id2idx, idx2label = misc.read_data(str(csv))
    assert id2idx == {7: 0, 3: 1} and idx2label == {0: "agama", 1: "drake"}
    assert "of" not in descr[0]["desc"] and "lizard" in descr[0]["desc"] and "," not in descr[0]["desc"]
    word_dict = misc.embed(word_dict, str(glove))
    descr = misc.cbow(descr, word_dict)
    np.testing.assert_allclose(descr[0]["cbow"].numpy(), [0.5, 0.5])         # mean over FOUND words only (misc.py:336)
    np.testing.assert_allclose(descr[1]["cbow"].numpy(), [1.0, 3.0])


def test_file_logger_format(tmp_path, capsys):
    p = tmp_path / "x.log"
    lg = misc.FileLogger(str(p))
    lg.Log("Starting epoch: 0")
    assert capsys.readouterr().err == "[1] Starting epoch: 0\n"              # misc.py:177
    assert re.match(r"^\d\d-\d\d-\d\d \d\d:\d\d:\d\d \[1\] Starting epoch: 0\n$", p.read_text())   # misc.py:183


def test_build_mask():
    m = misc.build_mask("0:3,5", 8)
    assert m.view(-1).tolist() == [1, 1, 1, 0, 0, 1, 0, 0]
    assert misc.build_mask("-1", 4).view(-1).tolist() == [0, 0, 0, 1]        # a single negative position (misc.py:398-400)


def test_clean_desc_and_embed_follow_the_reference_filters(tmp_path, monkeypatch):
    # the punctuation filter is a substring test on string.punctuation (misc.py:224): multi-character marks go too
    monkeypatch.setattr(misc, "word_tokenize", lambda text: text.split())
    assert misc.clean_desc("small () lizard <= of ./ warm :; regions ,-") == ["small", "lizard", "warm", "regions"]
    monkeypatch.undo()
    glove = tmp_path / "g.txt"
    glove.write_text("duck 1 1\nlizard 0 1\nduck 2 3\n")                    # a repeated word: the last row wins (misc.py:312-318)
    wd = misc.embed({"duck": {"id": 1}, "newt": {"id": 2}}, str(glove))
    assert wd["duck"]["emb"].tolist() == [2.0, 3.0] and wd["newt"]["emb"] is None


def test_hdf5_compound_roundtrip(tmp_path):
    """The record layouts of binary_vectors.py:24-46 (message dump)."""
    from multimodalgame_amd.binary_vectors import record_types
    comm_t, preds_t = record_types(6, 3)
    rs = np.random.RandomState(0)
    comm = np.zeros(5, comm_t)
    comm["ExampleId"] = [("img%d.jpg" % i).encode() for i in range(5)]
    comm["AgentId"], comm["Index"], comm["Target"], comm["Rank"] = b"S", np.arange(5), 2, [3, 1, 2, 1, 3]
    comm["BinaryProb"] = rs.rand(5, 6); comm["BinaryVec"] = (rs.rand(5, 6) < 0.5)
    preds = np.zeros(2, preds_t)
    preds["Predictions"] = rs.randn(2, 3); preds["StopProb"] = [[0.25], [0.75]]; preds["AgentId"] = b"R"
    p = str(tmp_path / "bv.hdf5")
    with hdf5io.File(p, "w") as f:
        f.write_struct("Communication", comm); f.write_struct("Predictions", preds)
    with hdf5io.File(p, "r") as f:
        c2, p2 = f.read_struct("Communication", comm_t), f.read_struct("Predictions", preds_t)
    for k in comm_t.names:
        np.testing.assert_array_equal(c2[k], comm[k])
    for k in preds_t.names:
        np.testing.assert_array_equal(p2[k], preds[k])


def test_host_shuffle_reproduces_random_shuffle():
    """libmmg's mmg_host_shuffle continues the interpreter's Mersenne-Twister state exactly as random.shuffle would
    (misc.py:270-271 order contract), for sizes around the rejection-sampling bit-length boundaries too."""
    import ctypes as C
    from multimodalgame_amd import _lib
    lib = _lib.load()
    for seed, n in ((11, 1), (12, 2), (11 + 7, 3000), (99, 4096), (5, 4097), (123, 65537)):
        random.seed(seed)
        st = random.getstate()[1]
        words = np.array(st[:-1], dtype=np.uint32)
        perm = np.arange(n, dtype=np.int64)
        assert lib.mmg_host_shuffle(words.ctypes.data_as(C.c_void_p), int(st[-1]), n, perm.ctypes.data_as(C.c_void_p)) == 0
        want = list(range(n))
        random.shuffle(want)
        assert perm.tolist() == want
    assert misc._shuffled_order(10) is not None and misc._FAST_SHUFFLE is True


def test_device_residency_is_decided_once_per_file(tmp_path, monkeypatch, capsys):
    """the resident-or-streaming decision of load_hdf5(device=...) was re-taken every epoch against the memory free
    at that moment -- after epoch 0 the resident copy and the gather blocks are allocated, so a mid-sized dataset flipped to host
    streaming from epoch 1 on.  Decided once per (file, device); memory torch's allocator merely caches counts as free; the
    streaming fallback says so on stderr."""
    p = tmp_path / "f.hdf5"
    p.write_bytes(b"x")
    data = {"avgpool_512": np.zeros((1000, 1, 512), np.float32)}          # 2 MB per copy
    state = {"free": 10 << 20}
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (state["free"], 64 << 20))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda device=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda device=None: 0)
    misc._RESIDENT_DECISION.clear()
    assert misc._fits_on_device(data, ("avgpool_512",), "cuda:0", str(p))            # 4 MB needed, 5 MB allowed
    state["free"] = 3 << 20                                                          # epoch 1: two copies are allocated now
    assert misc._fits_on_device(data, ("avgpool_512",), "cuda:0", str(p))            # ... the decision stands
    q = tmp_path / "g.hdf5"
    q.write_bytes(b"y")
    assert not misc._fits_on_device(data, ("avgpool_512",), "cuda:0", str(q))        # a NEW file is judged against today's memory
    assert "streaming" in capsys.readouterr().err
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda device=None: 8 << 20)  # cached-but-unused allocator blocks are free
    r = tmp_path / "h.hdf5"
    r.write_bytes(b"z")
    assert misc._fits_on_device(data, ("avgpool_512",), "cuda:0", str(r))
    misc._RESIDENT_DECISION.clear()
