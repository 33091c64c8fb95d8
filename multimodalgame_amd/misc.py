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


def clean_desc(desc):
    """Distinct lower-cased tokens of a description, first-occurrence order, without stop words and punctuation marks
    (the reference's filter chain, misc.py:220-226; it de-duplicates through set(), i.e. in hash order)."""
    seen, kept = set(), []
    for tok in word_tokenize(desc.lower()):
        if tok in seen:
            continue
        seen.add(tok)
        # `w not in string.punctuation` is a SUBSTRING test on the string (misc.py:224): it also drops multi-character
        # tokens such as "()" , "<=" or "./" that a tokenizer like nltk's can emit
        if tok not in STOPWORDS and tok not in string.punctuation:
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
            if head in word_dict:                      # a word listed twice: the LAST row wins, as in the reference's loop (misc.py:312-318)
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


def _file_key(hdf5_file):
    """Path + size + mtime: a file rewritten at the same path is a different dataset."""
    path = os.path.abspath(os.path.expanduser(hdf5_file))
    st = os.stat(path)
    return (path, st.st_size, st.st_mtime_ns)


def _dataset(hdf5_file, feats):
    key = (_file_key(hdf5_file), tuple(feats))
    if key not in _DATASET_CACHE:
        for old in [k for k in _DATASET_CACHE if k[0][0] == key[0][0] and k[1] == key[1]]:
            del _DATASET_CACHE[old]                      # the previous contents of this path
        with hdf5io.File(hdf5_file, "r") as f:
            d = {"Target": f.read("Target")}
            d["Location"] = f.read("Location") if "Location" in f else np.array([b""] * len(d["Target"]))
            for name in feats:
                d[name] = f.read(name)
        _DATASET_CACHE[key] = d
    return _DATASET_CACHE[key]


def _squeeze_feat(arr):
    """(N, 1, F) -> (N, F): the singleton axes of the stored features are dropped, the sample axis never is."""
    return arr.reshape(arr.shape[0], *[s for s in arr.shape[1:] if s != 1]) if arr.ndim > 2 else arr


_FAST_SHUFFLE = None          # None: not probed yet; True / False: libmmg's mmg_host_shuffle reproduces random.shuffle here


def _shuffled_order(n):
    """range(n) shuffled by Python's global generator AS random.shuffle WOULD (the state is read, the permutation is formed by
    libmmg's host helper, ~100x faster than the interpreter loop); falls back to random.shuffle itself if the library is
    missing or its result ever differed from the interpreter's on this Python build (probed once)."""
    global _FAST_SHUFFLE
    import ctypes as C

    def fast(m):
        from . import _lib
        st = random.getstate()[1]
        words = np.array(st[:-1], dtype=np.uint32)
        perm = np.arange(m, dtype=np.int64)
        _lib.check(_lib.load().mmg_host_shuffle(words.ctypes.data_as(C.c_void_p), int(st[-1]), m, perm.ctypes.data_as(C.c_void_p)))
        return perm
    if _FAST_SHUFFLE is None:
        saved = random.getstate()
        try:
            ok = True
            for seed, m in ((11, 1), (12, 2), (13, 1000), (14, 4099)):
                random.seed(seed)
                got = fast(m)
                want = list(range(m))
                random.shuffle(want)
                ok = ok and got.tolist() == want
            _FAST_SHUFFLE = ok
        except Exception:                               # noqa: BLE001  (no library in this process: the interpreter's own loop)
            _FAST_SHUFFLE = False
        finally:
            random.setstate(saved)                      # the probe never leaves the caller's generator reseeded
    # (the fast path reads the generator's state without advancing it; the epoch loop re-seeds before every shuffle --
    #  random.seed(11 + epoch), misc.py:270 -- so nothing downstream depends on the state a shuffle leaves behind)
    if _FAST_SHUFFLE:
        return fast(n)
    order = list(range(n))
    random.shuffle(order)
    return np.fromiter(order, dtype=np.int64, count=n)


def _epoch_order(dataset_size, batch_size, random_seed, shuffle, truncate_final_batch):
    """Per-batch index arrays in the reference's order (misc.py:262-282): random.seed(11 + epoch) shuffle of range(N)
    (Python's generator: the permutation is the contract), consecutive slices, indices SORTED inside a batch, last partial
    batch dropped unless truncate_final_batch.  The slicing / sorting is one numpy call over the whole epoch."""
    if shuffle:
        random.seed(11 + random_seed)
        arr = _shuffled_order(dataset_size)
    else:
        arr = np.arange(dataset_size, dtype=np.int64)
    n_full = dataset_size // batch_size
    full = np.sort(arr[:n_full * batch_size].reshape(n_full, batch_size), axis=1)
    batches = list(full)
    if truncate_final_batch and dataset_size - n_full * batch_size > 0:
        batches.append(np.sort(arr[n_full * batch_size:]))
    return batches


_RESIDENT_CACHE = {}


class _ResidentDataset(object):
    """One HDF5 file preloaded to `device`: the requested feature arrays as [N, F] float32 tensors and the
    class indices (map_labels applied ONCE per distinct raw label, not per sample per batch) as an int64 tensor."""

    def __init__(self, data, feats, device):
        self.device = device
        self.n = int(data["Target"].shape[0])
        self.location = data["Location"]
        self.raw_target = np.asarray(data["Target"]).astype(np.int64)
        self.feats = {name: torch.from_numpy(np.ascontiguousarray(_squeeze_feat(np.asarray(data[name], dtype=np.float32)))).to(device)
                      for name in feats}
        self._mapped = []                           # [(map_labels, device tensor)]: one entry per label map seen

    def targets(self, map_labels):
        for fn, t in self._mapped:
            if fn is map_labels:
                return t
        uniq = np.unique(self.raw_target)
        table = {int(u): map_labels(int(u)) for u in uniq}
        mapped = np.array([table[int(v)] for v in self.raw_target], dtype=np.int64)
        t = torch.from_numpy(mapped).to(self.device)
        self._mapped.append((map_labels, t))
        return t


def _resident(hdf5_file, feats, device):
    key = (_file_key(hdf5_file), tuple(feats), str(device))
    if key not in _RESIDENT_CACHE:
        for old in [k for k in _RESIDENT_CACHE if k[0][0] == key[0][0] and k[1:] == key[1:]]:
            del _RESIDENT_CACHE[old]                     # the previous contents of this path
        _RESIDENT_CACHE[key] = _ResidentDataset(_dataset(hdf5_file, feats), feats, device)
    return _RESIDENT_CACHE[key]


RESIDENT_FRACTION = 0.5       # of the device's free memory: resident copy + the epoch's gathered copy must fit below it


_RESIDENT_DECISION = {}       # (file key, feats, device) -> bool, decided ONCE per file and device


def _fits_on_device(data, feats, device, hdf5_file=None):
    """The device-resident epoch holds every requested feature array twice (the preloaded file and the epoch's batch-ordered
    gather).  A dataset beyond RESIDENT_FRACTION of the free device memory streams from the host instead, batch by batch, as
    the reference does (misc.py:284-302).  The decision is taken ONCE per (file, device): from epoch 1 on the resident copy is
    pinned in _RESIDENT_CACHE and the previous epoch's gather sits in torch's caching allocator, so the memory that is "free"
    then is two dataset sizes smaller than in epoch 0 -- re-deciding per epoch silently dropped mid-sized datasets to host
    streaming with the unused resident copy still allocated.  A file that is already resident only needs room for its gather."""
    device = torch.device(device)
    if device.type != "cuda":
        return True
    key = None
    if hdf5_file is not None:
        key = (_file_key(hdf5_file), tuple(feats), str(device))
        if key in _RESIDENT_DECISION:
            return _RESIDENT_DECISION[key]
    one = sum(int(np.prod(data[name].shape)) * 4 for name in feats)
    resident = key is not None and key in _RESIDENT_CACHE
    free, _ = torch.cuda.mem_get_info(device)
    free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)     # blocks torch's allocator holds but nobody uses
    fits = (one if resident else 2 * one) <= RESIDENT_FRACTION * free
    if key is not None:
        _RESIDENT_DECISION[key] = fits
    if not fits:
        import sys
        sys.stderr.write("load_hdf5: %s (%d MB per copy) does not fit twice into %.0f%% of the free memory of %s: streaming "
                         "batches from the host instead of the device-resident epoch loop\n"
                         % (hdf5_file, one >> 20, 100 * RESIDENT_FRACTION, device))
    return fits


class Epoch(object):
    """One epoch of a device-resident file in the reference's batch order (misc.py:257-302): `feats[name]` [n * B, F] and
    `target` [n * B] hold the samples of minibatch i in rows [i * B, (i + 1) * B) -- ONE gather per epoch -- so a minibatch is two
    tensor views and a run of consecutive minibatches is two pointers (include/mmg.h: mmg_train_steps).  Only whole batches
    (training drops the ragged last one, misc.py:279)."""

    def __init__(self, feats, target, batch):
        self.feats, self.target, self.B = feats, target, int(batch)
        self.n = int(target.size(0)) // self.B if self.B else 0

    def batch(self, i):
        lo, hi = i * self.B, (i + 1) * self.B
        out = {"target": self.target[lo:hi]}
        for name, t in self.feats.items():
            out[name] = t[lo:hi]
        return out


def load_epoch(hdf5_file, batch_size, random_seed, shuffle, map_labels=int, feats=("avgpool_512",), device=None, shard=None):
    """The training epoch as an Epoch (device-resident loop), or None when the file has to stream from the host
    (_fits_on_device) -- the caller then iterates load_hdf5.  Same order, same shard rule as load_hdf5."""
    if device is None:
        return None
    data = _dataset(hdf5_file, feats)
    if not _fits_on_device(data, feats, device, hdf5_file):
        return None
    batches = _epoch_order(int(data["Target"].shape[0]), batch_size, random_seed, shuffle, False)
    per = batch_size
    if shard is not None and shard[1] > 1:
        from .dist import shard_range
        lo, per = shard_range(batch_size, shard[0], shard[1])
        batches = [idx[lo:lo + per] for idx in batches]
    res = _resident(hdf5_file, feats, torch.device(device))
    perm = np.concatenate(batches) if batches else np.zeros(0, np.int64)
    if res.device.type == "cuda":
        # the permutation goes up from PINNED memory without blocking the host: a pageable copy waits, in stream order, for every
        # minibatch still queued -- the launch queue then runs dry once per epoch (46 minibatches at config 1) while the host forms
        # the gather, logs "Starting epoch" and enqueues the first run
        key = (str(res.device), perm.size)
        ring = _PERM_PINNED.setdefault(key, dict(buf=[torch.empty(perm.size, dtype=torch.int64, pin_memory=True) for _ in range(2)],
                                                 ev=[None, None], n=0))
        k = ring["n"] & 1                                # two buffers: the previous epoch's copy may still be queued
        ring["n"] += 1
        if ring["ev"][k] is not None:
            ring["ev"][k].synchronize()                  # (the copy of two epochs ago: long done unless the host ran far ahead)
        host = ring["buf"][k]
        host.numpy()[:] = perm
        flat = host.to(res.device, non_blocking=True)
        ring["ev"][k] = torch.cuda.Event()
        ring["ev"][k].record(torch.cuda.current_stream(res.device))
    else:
        flat = torch.from_numpy(perm).to(res.device)
    return Epoch({name: res.feats[name].index_select(0, flat) for name in feats}, res.targets(map_labels).index_select(0, flat), per)


_PERM_PINNED = {}


def load_hdf5(hdf5_file, batch_size, random_seed, shuffle, truncate_final_batch=False, map_labels=int,
              feats=("avgpool_512",), device=None, with_ids=True, shard=None):
    """Generator of batch dicts with the reference's order semantics (misc.py:257-302): random.seed(11 + epoch) shuffle of
    range(N), consecutive slices of the shuffled order, indices SORTED inside a batch, last partial batch
    dropped unless truncate_final_batch.  Only the requested feature datasets are read (the reference
    reads layer4_2 / avgpool_512 / fc for every batch even when unused).

    device given: the epoch loop is DEVICE-RESIDENT.  The file is preloaded to the device once (features + mapped targets), the
    epoch's permutation is built on the host once, and ONE gather per epoch lays the samples out in batch order; a batch is
    then two tensor views -- no per-batch numpy indexing, no per-sample Python, no host-to-device copy.
    with_ids=False skips the per-batch `example_ids` string array (the training loop never reads it).
    shard=(rank, world): data-parallel epoch loop -- the GLOBAL batch order above is formed on every rank and this rank keeps
    rows [rank * B / world, (rank + 1) * B / world) of every (sorted) batch (dist.shard_range), so the union over the ranks is
    exactly the single-process batch."""
    data = _dataset(hdf5_file, feats)
    batches = _epoch_order(int(data["Target"].shape[0]), batch_size, random_seed, shuffle, truncate_final_batch)
    if shard is not None and shard[1] > 1:
        from .dist import shard_range
        assert not truncate_final_batch, "a ragged final batch cannot be sharded evenly (training drops it, misc.py:279)"
        lo, per = shard_range(batch_size, shard[0], shard[1])
        batches = [idx[lo:lo + per] for idx in batches]
    if device is not None and not _fits_on_device(data, feats, device, hdf5_file):
        for idx in batches:                              # streaming fallback: host batches, one copy per batch
            batch = {"target": torch.tensor([map_labels(int(t)) for t in data["Target"][idx]], dtype=torch.int64).to(device)}
            if with_ids:
                batch["example_ids"] = data["Location"][idx]
            for name in feats:
                batch[name] = torch.from_numpy(_squeeze_feat(np.asarray(data[name][idx]))).float().to(device)
            yield batch
        return
    if device is not None:
        res = _resident(hdf5_file, feats, torch.device(device))
        flat = torch.from_numpy(np.concatenate(batches) if batches else np.zeros(0, np.int64)).to(res.device)
        target_ep = res.targets(map_labels).index_select(0, flat)
        feats_ep = {name: res.feats[name].index_select(0, flat) for name in feats}
        lo = 0
        for idx in batches:
            hi = lo + len(idx)
            batch = {"target": target_ep[lo:hi]}
            if with_ids:
                batch["example_ids"] = res.location[idx]
            for name in feats:
                batch[name] = feats_ep[name][lo:hi]
            lo = hi
            yield batch
        return
    for idx in batches:
        batch = {"target": torch.tensor([map_labels(int(t)) for t in data["Target"][idx]], dtype=torch.int64)}
        if with_ids:
            batch["example_ids"] = data["Location"][idx]
        for name in feats:
            batch[name] = torch.from_numpy(_squeeze_feat(np.asarray(data[name][idx]))).float()
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
        lo, sep, hi = piece.partition(":")
        if sep:
            mask[int(lo):int(hi)] = 1
        else:
            mask[int(lo)] = 1                          # a single position, negative ones included (misc.py:398-400)
    return mask
