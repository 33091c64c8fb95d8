"""Host utilities with the reference's names and formats (misc.py): checkpoint dict, FileLogger line
format, description pipeline (CSV -> tokens -> GloVe -> CBOW), HDF5 batch feed, init, bit-flip mask."""
import datetime
import itertools
import os
import random
import string
import sys

import numpy as np
import torch

from . import hdf5io


# ------------------------------------------------------------------ checkpoint (misc.py:42-92)
def recursively_set_device(inp, gpu):
    if hasattr(inp, "keys"):
        for k in inp.keys():
            inp[k] = recursively_set_device(inp[k], gpu)
    elif isinstance(inp, list):
        return [recursively_set_device(ii, gpu) for ii in inp]
    elif isinstance(inp, tuple):
        return tuple(recursively_set_device(ii, gpu) for ii in inp)
    elif hasattr(inp, "cpu"):
        inp = inp.cuda() if gpu >= 0 else inp.cpu()
    return inp


def torch_save(filename, data, models_dict, optimizers_dict, gpu=-1):
    """Same file layout as misc.py:58-69: {'data', 'optimizers', 'models'}, tensors on the CPU."""
    models_to_save = {k: recursively_set_device({kk: vv.detach().clone() for kk, vv in v.state_dict().items()}, gpu=-1)
                      for k, v in models_dict.items()}
    optimizers_to_save = {k: recursively_set_device(v.state_dict(), gpu=-1) for k, v in optimizers_dict.items()}
    torch.save({"data": data, "optimizers": optimizers_to_save, "models": models_to_save}, filename)


def torch_load(filename, models_dict, optimizers_dict):
    filename = os.path.expanduser(filename)
    if not os.path.exists(filename):
        raise Exception("File does not exist: " + filename)                # misc.py:81-82
    checkpoint = torch.load(filename, map_location="cpu", weights_only=False)
    for k, v in models_dict.items():
        v.load_state_dict(checkpoint["models"][k])
    for k, v in optimizers_dict.items():
        v.load_state_dict(checkpoint["optimizers"][k])
    return checkpoint["data"]


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


def clean_desc(desc):
    words = word_tokenize(desc.lower())
    words = list(dict.fromkeys(words))                                       # remove duplicates (stable order)
    words = [w for w in words if w not in STOPWORDS]
    words = [w for w in words if w not in string.punctuation]
    return words


def read_data(input_descr):
    """misc.py:229-254: label_id,label,free text (text may contain commas)."""
    descr, word_dict, dict_size, num_descr = {}, {}, 0, 0
    label_id_to_idx, idx_to_label = {}, {}
    with open(input_descr, "r") as f:
        for i, line in enumerate(f):
            line = line.strip()
            parts = line.split(",")
            label_id, label = parts[:2]
            desc = clean_desc(line[len(label_id) + len(label) + 2:])
            for w in desc:
                if w not in word_dict:
                    dict_size += 1
                    word_dict[w] = {"id": dict_size}
            descr[num_descr] = {"name": label, "desc": desc}
            num_descr += 1
            label_id_to_idx[int(label_id)] = i
            idx_to_label[i] = label
    return descr, word_dict, dict_size, label_id_to_idx, idx_to_label


def embed(word_dict, emb):
    glove = {}
    with open(emb, "r") as f:
        for line in f:
            word = line.strip().split(" ")
            if word[0] in word_dict:
                glove[word[0]] = torch.tensor([float(s) for s in word[1:]])
    for k in word_dict:
        word_dict[k]["emb"] = glove.get(k, None)
    return word_dict


def cbow(descr, word_dict):
    """Mean of the GloVe vectors found; missing words are zero rows and do not count (misc.py:324-340)."""
    emb_size = next(len(v["emb"]) for v in word_dict.values() if v["emb"] is not None)
    for mammal in descr:
        num_w = 0
        desc_set = torch.zeros(len(descr[mammal]["desc"]), emb_size)
        for i_w, w in enumerate(descr[mammal]["desc"]):
            if word_dict[w]["emb"] is not None:
                desc_set[i_w] = word_dict[w]["emb"]
                num_w += 1
        desc_cbow = desc_set.sum(0)
        if num_w > 0:
            desc_cbow = desc_cbow / num_w
        descr[mammal]["cbow"] = desc_cbow
        descr[mammal]["set"] = desc_set
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
def xavier_normal(tensor, gain=1):
    fan_out, fan_in = tensor.size(0), tensor.size(1)
    std = gain * np.sqrt(2.0 / (fan_in + fan_out))
    with torch.no_grad():
        return tensor.normal_(0, std)


def build_mask(region_str, size):
    regions = [r.split(":") for r in region_str.split(",")]
    regions = [[int(r[0])] if len(r) == 1 else list(range(int(r[0]), int(r[1]))) for r in regions]
    index = torch.LongTensor(list(itertools.chain(*regions)))
    mask = torch.zeros(size, 1)
    mask[index] = 1
    return mask
