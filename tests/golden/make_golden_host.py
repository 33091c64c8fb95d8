#!/usr/bin/env python
"""Golden vectors for the host-side rows f3 / f4 of SURVEY.md §8, produced by running the REFERENCE's own functions.

  g6_desc_pipeline.npz    misc.read_data -> misc.embed -> misc.cbow (misc.py:220-254, 306-340) on the committed mini CSV /
                          GloVe snippet (tests/golden/fixtures/), with nltk's word_tokenize / stop-word list replaced by the
                          repo's own (multimodalgame_amd.misc) -- nltk is not installable here, so what is pinned is
                          everything AROUND the tokenizer: duplicate removal, stop-word / punctuation filtering, the
                          label_id -> row map, the comma-tolerant CSV split, the missing-word rule of the CBOW mean.
  g7_binary_vectors.npz   binary_vectors.extract_binary (binary_vectors.py:12-135) for one deterministic dev batch of the
                          reference agents (weights / inputs from the seeded fillers of oracle/cpu_ref.py), h5py replaced by a
                          recorder: the `Communication` / `Predictions` records incl. the Rank formula (:98) and the
                          Index convention (sender 2i, receiver 2i + 1).

  g8_eval_dev.npz         model.eval_dev (model.py:580-722) itself, on two deterministic dev batches (the second one SHORT: the
                          nominal-batch denominator of :667), load_hdf5 replaced by a generator of the seeded batches:
                          accuracy, conversation-length mean / std, both mean Hamming distances and the confusion matrix
                          (sklearn.metrics.confusion_matrix: labels = the sorted classes that OCCUR in truth or prediction).

Run in the build container only (needs /root/reference); only numbers are written.
usage: python tests/golden/make_golden_host.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from multimodalgame_amd import misc as own_misc  # noqa: E402  (tokenizer + stop-word list only)

FIX = os.path.join(HERE, "fixtures")


def load_ref_misc(ref_dir):
    """The reference's misc.py as text, one py3 substitution (dict.values()[0], misc.py:326), tokenizer stubs."""
    src = open(os.path.join(ref_dir, "misc.py")).read()
    src = MG._sub(src, "emb_size = len(word_dict.values()[0][\"emb\"])",
                  "emb_size = len([v for v in word_dict.values() if v[\"emb\"] is not None][0][\"emb\"])")
    mod = types.ModuleType("reference_misc")
    mod.__file__ = os.path.join(ref_dir, "misc.py")
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    mod.word_tokenize = own_misc.word_tokenize
    mod.stopwords = types.SimpleNamespace(words=lambda lang: sorted(own_misc.STOPWORDS))
    return mod


def case_g6(ref_dir):
    m = load_ref_misc(ref_dir)
    csv, glove = os.path.join(FIX, "descriptions_mini.csv"), os.path.join(FIX, "glove_mini.8d.txt")
    descr, word_dict, dict_size, label_id_to_idx, idx_to_label = m.read_data(csv)
    word_dict = m.embed(word_dict, glove)
    descr = m.cbow(descr, word_dict)
    n = len(descr)
    out = dict(n_classes=n, dict_size=dict_size,
               cbow=np.stack([descr[i]["cbow"].numpy() for i in range(n)]),
               names=np.array([descr[i]["name"] for i in range(n)]),
               # the reference removes duplicates through set(): word ORDER is hash order -- pin the sets
               desc_sorted=np.array(["|".join(sorted(descr[i]["desc"])) for i in range(n)]),
               set_rowsum=np.stack([descr[i]["set"].sum(0).numpy() for i in range(n)]),
               label_ids=np.array(sorted(label_id_to_idx.keys())),
               label_rows=np.array([label_id_to_idx[k] for k in sorted(label_id_to_idx.keys())]),
               idx_to_label=np.array([idx_to_label[i] for i in range(n)]),
               vocab_sorted=np.array(sorted(word_dict.keys())),
               vocab_found=np.array([int(word_dict[w]["emb"] is not None) for w in sorted(word_dict.keys())]))
    return out


class _RecDataset(object):
    """h5py.Dataset stand-in: resize + tail slice assignment from an iterable of record tuples."""

    def __init__(self, dtype):
        self.dtype, self.rows = dtype, []

    @property
    def shape(self):
        return (len(self.rows),)

    def resize(self, n, axis=0):
        self._pending = n - len(self.rows)

    def __setitem__(self, key, value):
        new = list(value)
        assert len(new) == self._pending
        self.rows += new


class _RecFile(object):
    last = None

    def __init__(self, path, mode):
        self.ds = {}
        _RecFile.last = self

    def create_dataset(self, name, shape, maxshape=None, dtype=None):
        self.ds[name] = _RecDataset(dtype)
        return self.ds[name]


def case_g7(ref, FLAGS, ref_dir):
    fl = MG.make_flags(use_binary=True, fixed_exchange=False, max_exchange=4, batch_size=6, top_k_train=2, **MG.TINY)
    MG.set_flags(FLAGS, fl)
    FLAGS.binary_output = "unused"
    FLAGS.data_context = "fc"
    torch.manual_seed(0)
    models = MG.build_ref_models(ref, FLAGS)
    seeds = dict(weights=51, data=52)
    cpu_ref.load_filled(models, seed=seeds["weights"])
    with torch.no_grad():
        models["receiver"].s.bias.fill_(1.2)                           # conversations of several steps (round(prod p_s))
    n_classes, batch = 5, 6
    x, _, desc = cpu_ref.synthetic_batch(batch, n_classes, fl.img_feat_dim, fl.wv_dim, seed=seeds["data"])
    target = np.full((batch,), 3, dtype=np.int64)                      # "Rank only works if there is one target"
    ids = np.array([("img_%03d.jpg" % i).encode() for i in range(batch)])

    def fake_load_hdf5(dev_file, batch_size, epoch, shuffle, truncate_final_batch=False, map_labels=int):
        yield {"target": torch.from_numpy(target), "avgpool_512": torch.from_numpy(x), "example_ids": ids}

    src = open(os.path.join(ref_dir, "binary_vectors.py")).read()
    mod = types.ModuleType("reference_binary_vectors")
    mod.__file__ = os.path.join(ref_dir, "binary_vectors.py")
    sys.modules["h5py"].File = _RecFile
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    for m in models.values():
        m.eval()
    mod.extract_binary(FLAGS, fake_load_hdf5, ref.exchange, "dev", batch, 0, False, False, 2, models["sender"],
                       models["receiver"], {"desc": torch.from_numpy(desc)}, int, "unused")
    f = _RecFile.last
    comm, preds = f.ds["Communication"].rows, f.ds["Predictions"].rows
    out = dict(seed_weights=seeds["weights"], seed_data=seeds["data"], n_classes=n_classes, batch=batch, target=target,
               example_ids=ids,
               comm_agent=np.array([r[1] for r in comm]), comm_index=np.array([r[2] for r in comm], dtype=np.int32),
               comm_target=np.array([r[3] for r in comm], dtype=np.int32), comm_rank=np.array([r[4] for r in comm], dtype=np.int32),
               comm_prob=np.stack([np.asarray(r[5], np.float32) for r in comm]),
               comm_vec=np.stack([np.asarray(r[6], np.float32) for r in comm]),
               comm_ids=np.array([r[0] for r in comm]),
               pred_index=np.array([r[2] for r in preds], dtype=np.int32), pred_rank=np.array([r[4] for r in preds], dtype=np.int32),
               pred_scores=np.stack([np.asarray(r[5], np.float32) for r in preds]),
               pred_stop_prob=np.stack([np.asarray(r[6], np.float32).reshape(1) for r in preds]),
               pred_stop_vec=np.stack([np.asarray(r[7], np.float32).reshape(1) for r in preds]),
               pred_stop_mask=np.stack([np.asarray(r[8], np.float32).reshape(1) for r in preds]))
    out["meta"] = MG.flags_to_meta(fl, n_classes, batch, dict(weights=51, data=52, uniforms=0), 1)
    return out


def case_g8(ref, FLAGS, tmp_dir):
    fl = MG.make_flags(use_binary=True, fixed_exchange=False, max_exchange=5, batch_size=8, top_k_train=2, top_k_dev=2, **MG.TINY)
    MG.set_flags(FLAGS, fl)
    FLAGS.conf_mat = os.path.join(tmp_dir, "g8.conf_mat.txt")
    FLAGS.attn_extra_context, FLAGS.bit_flip, FLAGS.corrupt_region = False, False, None
    torch.manual_seed(0)
    models = MG.build_ref_models(ref, FLAGS)
    seeds = dict(weights=79, data=62)
    cpu_ref.load_filled(models, seed=seeds["weights"])
    with torch.no_grad():
        models["receiver"].s.bias.fill_(0.9)                           # conversations of mixed lengths (round(prod p_s))
    n_classes, batch, sizes = 7, 8, (8, 5)                             # nominal dev batch 8, the final one holds 5 samples
    x0, t0, desc = cpu_ref.synthetic_batch(sizes[0], n_classes, fl.img_feat_dim, fl.wv_dim, seed=seeds["data"])
    x1, t1, _ = cpu_ref.synthetic_batch(sizes[1], n_classes, fl.img_feat_dim, fl.wv_dim, seed=seeds["data"] + 1)
    t0, t1 = np.minimum(t0, 5), np.minimum(t1, 5)                      # class 6 never occurs as a target: confusion-matrix labels

    def fake_load_hdf5(dev_file, batch_size, epoch, shuffle, truncate_final_batch=False, map_labels=int):
        assert truncate_final_batch and batch_size == batch
        for x, t in ((x0, t0), (x1, t1)):
            yield {"target": torch.from_numpy(t), "avgpool_512": torch.from_numpy(x)}
    ref.load_hdf5 = fake_load_hdf5
    for m in models.values():
        m.eval()
    acc, extra = ref.eval_dev("dev", batch, 0, False, False, fl.top_k_dev, models["sender"], models["receiver"],
                              {"desc": torch.from_numpy(desc)}, int, "unused")
    conf = np.loadtxt(FLAGS.conf_mat, delimiter=",", dtype=np.int64, ndmin=2)
    out = dict(seed_weights=seeds["weights"], seed_data=seeds["data"], n_classes=n_classes, batch=batch, sizes=np.array(sizes),
               target0=t0, target1=t1, accuracy=float(acc), conversation_lengths_mean=float(extra["conversation_lengths_mean"]),
               conversation_lengths_std=float(extra["conversation_lengths_std"]),
               hamming_sen_mean=float(extra["hamming_sen_mean"]), hamming_rec_mean=float(extra["hamming_rec_mean"]),
               conf_mat=conf, s_bias=0.9)
    out["meta"] = MG.flags_to_meta(fl, n_classes, batch, dict(weights=79, data=62, uniforms=0), 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    ref, FLAGS = MG.load_reference(a.ref)
    np.savez_compressed(os.path.join(a.out, "g6_desc_pipeline.npz"), **case_g6(a.ref))
    np.savez_compressed(os.path.join(a.out, "g7_binary_vectors.npz"), **case_g7(ref, FLAGS, a.ref))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        np.savez_compressed(os.path.join(a.out, "g8_eval_dev.npz"), **case_g8(ref, FLAGS, tmp))
    print("wrote g6_desc_pipeline.npz, g7_binary_vectors.npz, g8_eval_dev.npz")


if __name__ == "__main__":
    main()
