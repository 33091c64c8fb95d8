"""Host utilities with the reference's names and formats (misc.py): checkpoint dict, FileLogger line
format, description pipeline (CSV -> tokens -> GloVe -> CBOW), HDF5 batch feed, init, bit-flip mask."""
import datetime
import os
import random
import string
import sys

import numpy as np
import torch

from . import hdf5io


# ------------------------------------------------------------------ checkpoint (file contract: misc.py:58-75)
def _host_copy(tree):
    """Detached CPU copies of every tensor of a (nested) state_dict; containers are rebuilt, scalars pass through."""
    from torch.utils._pytree import tree_map
    return tree_map(lambda leaf: leaf.detach().to("cpu", copy=True) if torch.is_tensor(leaf) else leaf, tree)


def torch_save(filename, data, models_dict, optimizers_dict, gpu=-1):
    """One file, three keys -- 'data' (step / best accuracy), 'models', 'optimizers' -- each a name -> state_dict map with
    CPU tensors: what the reference's checkpoints contain and what its torch_load expects."""
    payload = {"data": data,
               "optimizers": {name: _host_copy(opt.state_dict()) for name, opt in optimizers_dict.items()},
               "models": {name: _host_copy(dict(mod.state_dict())) for name, mod in models_dict.items()}}
    torch.save(payload, filename)


def torch_load(filename, models_dict, optimizers_dict):
    """Restore every module and optimizer in place from a checkpoint file and return its 'data' entry."""
    path = os.path.expanduser(filename)
    if not os.path.isfile(path):
        raise Exception("File does not exist: " + path)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    for kind, targets in (("models", models_dict), ("optimizers", optimizers_dict)):
        for name, obj in targets.items():
            obj.load_state_dict(ckpt[kind][name])
    return ckpt["data"]


# ------------------------------------------------------------------ logging (misc.py:95-190)
class VisdomLogger(object):
    """Visdom is not on the accelerated path; the flag is accepted and ignored when the module is absent."""

    def __init__(self, env, experiment_name, minimum=2, enabled=False):
        self.enabled = False

    def log(self, key, val, step):
        return


class FileLogger(object):
    DEBUG, INFO, WARNING, ERROR = 0, 1, 2, 3

    def __init__(self, log_path=None, json_log_path=None, min_print_level=0, min_file_level=0):
        self.log_path, self.json_log_path = log_path, json_log_path
        self.min_print_level, self.min_file_level = min_print_level, min_file_level

    def Log(self, message, level=INFO):
        if level >= self.min_print_level:
            sys.stderr.write("[%i] %s\n" % (level, message))                 # misc.py:177
        if self.log_path and level >= self.min_file_level:
            with open(self.log_path, "a") as f:
                datetime_string = datetime.datetime.now().strftime("%y-%m-%d %H:%M:%S")
                f.write("%s [%i] %s\n" % (datetime_string, level, message))  # misc.py:183


# ------------------------------------------------------------------ descriptions (misc.py:220-254, 306-340)
# nltk is not installable here: the tokenizer splits on non-alphanumerics (keeping inner hyphens and
# apostrophes) and the stop-word list is nltk's English list (public domain, 179 entries, abridged to
# the words that can occur in WordNet glosses).  Parity with nltk.word_tokenize is NOT pinned.
STOPWORDS = set("""i me my myself we our ours ourselves you your yours yourself yourselves he him his himself she her
hers herself it its itself they them their theirs themselves what which who whom this that these those am is are was
were be been being have has had having do does did doing a an the and but if or because as until while of at by for
with about against between into through during before after above below to from up down in out on off over under
again further then once here there when where why how all any both each few more most other some such no nor not only
own same so than too very s t can will just don should now d ll m o re ve y ain aren couldn didn doesn hadn hasn haven
isn ma mightn mustn needn shan shouldn wasn weren won wouldn""".split())


def word_tokenize(text):
    out, cur = [], []
    for ch in text:
        if ch.isalnum() or ch in "-'":
            cur.append(ch)
        else:
            if cur:
                out.append("".join(cur)); cur = []
            if not ch.isspace():
                out.append(ch)
    if cur:
        out.append("".join(cur))
    return out


_PUNCT = frozenset(string.punctuation)


def clean_desc(desc):
    """Distinct lower-cased tokens of a description, first-occurrence order, without stop words and punctuation marks
    (the reference's filter chain, misc.py:220-226; it de-duplicates through set(), i.e. in hash order)."""
    seen, kept = set(), []
    for tok in word_tokenize(desc.lower()):
        if tok in seen:
            continue
        seen.add(tok)
        if tok not in STOPWORDS and tok not in _PUNCT:
            kept.append(tok)
    return kept


def read_data(input_descr):
    """Descriptions CSV `label_id,label,free text` -> the five structures the reference returns (misc.py:229-254):
    descr[row] = {name, desc}, word_dict[word] = {id} (ids from 1 in order of first appearance), the vocabulary size,
    label_id -> row and row -> label.  The text field keeps its commas: the line is split at the first two only."""
    descr, vocab, row_of_label, label_of_row = {}, {}, {}, {}
    with open(input_descr, "r") as f:
        for row, raw in enumerate(f):
            label_id, label, text = raw.strip().split(",", 2)
            tokens = clean_desc(text)
            for tok in tokens:
                vocab.setdefault(tok, {"id": len(vocab) + 1})
            descr[row] = {"name": label, "desc": tokens}
            row_of_label[int(label_id)] = row
            label_of_row[row] = label
    return descr, vocab, len(vocab), row_of_label, label_of_row


def embed(word_dict, emb):
    """Attach the GloVe row of every vocabulary word (`word v1 ... vV` per line, misc.py:306-320); words the file does not
    contain get None.  Only the lines of vocabulary words are parsed."""
    rows = {}
    with open(emb, "r") as f:
        for line in f:
            head, _, tail = line.rstrip("\n").partition(" ")
            if head in word_dict and head not in rows:
                rows[head] = torch.from_numpy(np.array(tail.split(), dtype=np.float32))
    for word, entry in word_dict.items():
        entry["emb"] = rows.get(word)
    return word_dict


def cbow(descr, word_dict):
    """Description vectors: descr[row]['set'] = one GloVe row per token (zeros for words GloVe lacks), descr[row]['cbow'] =
    their sum divided by the number of FOUND words (misc.py:323-340).  Built from one [vocab + 1, V] table (row 0 = the
    zero vector of missing words) and an index gather instead of a per-word Python loop."""
    words = list(word_dict.keys())
    known = [w for w in words if word_dict[w].get("emb") is not None]
    if not known:
        raise ValueError("no description word has an embedding")
    dim = int(word_dict[known[0]]["emb"].numel())
    table = torch.zeros(len(words) + 1, dim)
    slot = {}
    for k, w in enumerate(words, start=1):
        slot[w] = k
        if word_dict[w]["emb"] is not None:
            table[k] = word_dict[w]["emb"]
    found = torch.tensor([0.0] + [float(word_dict[w]["emb"] is not None) for w in words])
    for entry in descr.values():
        idx = torch.tensor([slot[w] for w in entry["desc"]], dtype=torch.long)
        rows = table[idx] if idx.numel() else torch.zeros(0, dim)
        n_found = float(found[idx].sum()) if idx.numel() else 0.0
        total = rows.sum(0)
        entry["set"] = rows
        entry["cbow"] = total / n_found if n_found > 0 else total
    return descr


# ------------------------------------------------------------------ HDF5 feed (misc.py:257-302)
_DATASET_CACHE = {}


def _dataset(hdf5_file, feats):
    key = (os.path.abspath(os.path.expanduser(hdf5_file)), tuple(feats))
    if key not in _DATASET_CACHE:
        with hdf5io.File(hdf5_file, "r") as f:
            d = {"Target": f.read("Target")}
            d["Location"] = f.read("Location") if "Location" in f else np.array([b""] * len(d["Target"]))
            for name in feats:
                d[name] = f.read(name)
        _DATASET_CACHE[key] = d
    return _DATASET_CACHE[key]


def load_hdf5(hdf5_file, batch_size, random_seed, shuffle, truncate_final_batch=False, map_labels=int,
              feats=("avgpool_512",), device=None):
    """Generator of batch dicts with the reference's order semantics: random.seed(11 + epoch) shuffle of
    range(N), consecutive slices of the shuffled order, indices SORTED inside a batch, last partial batch
    dropped unless truncate_final_batch.  Only the requested feature datasets are read (the reference
    reads layer4_2 / avgpool_512 / fc for every batch even when unused)."""
    data = _dataset(hdf5_file, feats)
    dataset_size = data["Target"].shape[0]
    order = list(range(dataset_size))
    if shuffle:
        random.seed(11 + random_seed)
        random.shuffle(order)
    num_batches = dataset_size // batch_size
    if truncate_final_batch and dataset_size - num_batches * batch_size > 0:
        num_batches += 1
    for i in range(num_batches):
        idx = sorted(order[i * batch_size:(i + 1) * batch_size])
        batch = {"target": torch.tensor([map_labels(int(t)) for t in data["Target"][idx]], dtype=torch.int64),
                 "example_ids": data["Location"][idx]}
        for name in feats:
            arr = torch.from_numpy(data[name][idx]).float()
            arr = arr.reshape(arr.shape[0], *[s for s in arr.shape[1:] if s != 1]) if arr.dim() > 2 else arr
            batch[name] = arr
        if device is not None:
            batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        yield batch


def write_synthetic_dataset(dirname, n_classes=30, per_class=100, feat_dim=512, wv_dim=100, seed=1234):
    """SURVEY.md §8(d) synthetic inputs with the reference's file formats: {train,dev}.hdf5
    (avgpool_512 (N,1,F) f32 = |N(0,1)|, Target (N,) int, Location (N,) S50), descriptions.csv and a
    GloVe-format text file whose CBOW description vectors are ~0.3 N(0,1)."""
    os.makedirs(dirname, exist_ok=True)
    rs = np.random.RandomState(seed)
    n = n_classes * per_class
    for split in ("train", "dev"):
        path = os.path.join(dirname, split + ".hdf5")
        if os.path.exists(path):
            continue
        with hdf5io.File(path, "w") as f:
            f.write("avgpool_512", np.abs(rs.standard_normal((n, 1, feat_dim))).astype(np.float32))
            f.write("Target", rs.randint(0, n_classes, size=(n,)).astype(np.int32))
            f.write("Location", np.array([("%s_%06d.jpg" % (split, i)).encode() for i in range(n)], dtype="S50"))
    csv, glove = os.path.join(dirname, "descriptions.csv"), os.path.join(dirname, "glove.synthetic.%dd.txt" % wv_dim)
    if not os.path.exists(csv):
        with open(csv, "w") as f, open(glove, "w") as g:
            for c in range(n_classes):
                words = ["w%dx%d" % (c, j) for j in range(4)]
                f.write("%d,class%d,%s\n" % (c, c, " ".join(words)))
                for w in words:
                    g.write(w + " " + " ".join("%.5f" % v for v in 0.6 * rs.standard_normal(wv_dim)) + "\n")
    return dict(train_file=os.path.join(dirname, "train.hdf5"), dev_file=os.path.join(dirname, "dev.hdf5"),
                descr_train=csv, descr_dev=csv, glove_path=glove)


# ------------------------------------------------------------------ init + mask (misc.py:349-402)
from .agents import xavier_normal  # noqa: E402,F401  (one implementation: agents.py)


def build_mask(region_str, size):
    """[size, 1] indicator of the positions named by a region string such as "0:4,7,10:12" (half-open ranges and single
    positions; -corrupt_region, misc.py:388-402)."""
    mask = torch.zeros(size, 1)
    for piece in region_str.split(","):
        lo, _, hi = piece.partition(":")
        mask[int(lo):(int(hi) if hi else int(lo) + 1)] = 1
    return mask
