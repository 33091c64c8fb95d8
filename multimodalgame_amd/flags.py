"""A python-gflags-compatible flag registry (python-gflags is not installable here), with the
reference's 74 flags (model.py:1639-1741), its four presets (model.py:1605-1636) and the derived
defaults of default_flags() (model.py:1744-1810).

Syntax kept from gflags: single- or double-dash names, ``-name value`` / ``-name=value``,
booleans as ``-name`` / ``-noname`` / ``-name=false``, enum validation, later occurrences win,
``FLAGS(argv)`` may be called repeatedly (the reference re-parses argv after applying a preset so
that the command line overrides it, model.py:1750-1754).
"""
import json
import os
import sys
import time


class FlagsError(Exception):
    pass


class _Flag(object):
    def __init__(self, name, default, kind, choices=None, help=""):
        self.name, self.default, self.kind, self.choices, self.help = name, default, kind, choices, help
        self.value = default

    def parse(self, text):
        if self.kind == "string":
            return text
        if self.kind == "integer":
            return int(text)
        if self.kind == "float":
            return float(text)
        if self.kind == "boolean":
            t = text.lower()
            if t in ("true", "t", "1", "yes", "y"):
                return True
            if t in ("false", "f", "0", "no", "n"):
                return False
            raise FlagsError("flag -%s: bad boolean %r" % (self.name, text))
        if self.kind == "enum":
            if text not in self.choices:
                raise FlagsError("flag -%s: value should be one of <%s>" % (self.name, "|".join(self.choices)))
            return text
        raise AssertionError(self.kind)


class FlagValues(object):
    def __init__(self):
        object.__setattr__(self, "_flags", {})

    def _define(self, flag):
        self._flags[flag.name] = flag

    def __getattr__(self, name):
        fl = object.__getattribute__(self, "_flags")
        if name in fl:
            return fl[name].value
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name not in self._flags:
            raise AttributeError("unknown flag " + name)
        self._flags[name].value = value

    def FlagValuesDict(self):
        return {k: f.value for k, f in self._flags.items()}

    def Reset(self):
        for f in self._flags.values():
            f.value = f.default

    def __call__(self, argv):
        """Parse argv[1:]; returns [argv[0]] + the non-flag arguments (gflags behaviour)."""
        rest = [argv[0]] if argv else []
        i = 1
        while i < len(argv):
            arg = argv[i]
            i += 1
            if arg == "--":
                rest += argv[i:]
                break
            if not arg.startswith("-") or arg == "-":
                rest.append(arg)
                continue
            body = arg.lstrip("-")
            name, eq, val = body.partition("=")
            if name in self._flags:
                flag = self._flags[name]
                if flag.kind == "boolean":
                    flag.value = flag.parse(val) if eq else True
                    continue
                if not eq:
                    if i >= len(argv):
                        raise FlagsError("flag -%s needs a value" % name)
                    val = argv[i]
                    i += 1
                flag.value = flag.parse(val)
            elif name.startswith("no") and name[2:] in self._flags and self._flags[name[2:]].kind == "boolean":
                if eq:
                    raise FlagsError("flag -%s takes no value" % name)
                self._flags[name[2:]].value = False
            else:
                raise FlagsError("Unknown command line flag '%s'" % name)
        return rest


FLAGS = FlagValues()


def DEFINE_string(name, default, help=""):
    FLAGS._define(_Flag(name, default, "string", help=help))


def DEFINE_boolean(name, default, help=""):
    FLAGS._define(_Flag(name, default, "boolean", help=help))


def DEFINE_integer(name, default, help=""):
    FLAGS._define(_Flag(name, default, "integer", help=help))


def DEFINE_float(name, default, help=""):
    FLAGS._define(_Flag(name, default, "float", help=help))


def DEFINE_enum(name, default, choices, help=""):
    FLAGS._define(_Flag(name, default, "enum", choices=list(choices), help=help))


def define_flags():
    """Every flag of model.py:1639-1741 with its default, plus additive ones this build needs."""
    if "max_exchange" in FLAGS._flags:
        return
    # Debug settings
    DEFINE_string("branch", None); DEFINE_string("sha", None); DEFINE_boolean("debug", False)
    # Convenience settings
    DEFINE_integer("save_after", 1000); DEFINE_integer("save_interval", 100); DEFINE_string("checkpoint", None)
    DEFINE_string("conf_mat", None); DEFINE_string("log_path", "./logs"); DEFINE_string("log_file", None)
    DEFINE_string("eval_csv_file", None); DEFINE_string("json_file", None); DEFINE_string("log_load", None)
    DEFINE_boolean("eval_only", False)
    # Extract settings
    DEFINE_boolean("binary_only", False); DEFINE_string("binary_output", None)
    # Performance settings
    DEFINE_boolean("cuda", False)
    # Display settings
    DEFINE_string("env", "main"); DEFINE_boolean("visdom", False); DEFINE_boolean("use_alpha", False)
    DEFINE_string("experiment_name", None); DEFINE_integer("log_interval", 50); DEFINE_integer("log_dev", 1000)
    # Data settings
    DEFINE_enum("wv_type", "glove.6B", ["fake", "glove.6B", "none"]); DEFINE_integer("wv_dim", 100)
    DEFINE_string("descr_train", "descriptions.csv"); DEFINE_string("descr_dev", "descriptions.csv")
    DEFINE_string("train_file", "train.hdf5"); DEFINE_string("dev_file", "dev.hdf5")
    DEFINE_enum("images", "mammal", ["cifar", "mammal"])
    DEFINE_string("glove_path", "./glove.6B/glove.6B.100d.txt")
    DEFINE_boolean("shuffle_train", True); DEFINE_boolean("shuffle_dev", False)
    # Model settings
    DEFINE_enum("model_type", None, ["Fixed", "Adaptive", "FixedAttention", "AdaptiveAttention"])
    DEFINE_enum("img_feat", "avgpool_512", ["layer4_2", "avgpool_512", "fc"])
    DEFINE_enum("data_context", "fc", ["fc"]); DEFINE_enum("sender_mix", "sum", ["sum", "prod", "mou"])
    DEFINE_integer("img_feat_dim", 4096); DEFINE_integer("img_h_dim", 100); DEFINE_integer("baseline_hid_dim", 500)
    DEFINE_integer("sender_out_dim", 50); DEFINE_integer("rec_hidden", 128); DEFINE_integer("rec_out_dim", 1)
    DEFINE_integer("rec_w_dim", 50); DEFINE_integer("rec_s_dim", 1); DEFINE_boolean("use_binary", True)
    DEFINE_boolean("ignore_receiver", False); DEFINE_boolean("ignore_code", False); DEFINE_boolean("block_y", True)
    DEFINE_float("first_rec", 0); DEFINE_float("flipout_rec", None); DEFINE_float("flipout_sen", None)
    DEFINE_boolean("flipout_dev", False); DEFINE_boolean("s_prob_prod", True); DEFINE_boolean("visual_attn", False)
    DEFINE_integer("attn_dim", 256); DEFINE_boolean("attn_extra_context", False); DEFINE_integer("attn_context_dim", 4096)
    DEFINE_boolean("desc_attn", False); DEFINE_integer("desc_attn_dim", 64)
    DEFINE_integer("top_k_dev", 6); DEFINE_integer("top_k_train", 6)
    # Optimization settings
    DEFINE_enum("optim_type", "RMSprop", ["Adam", "SGD", "RMSprop"]); DEFINE_integer("batch_size", 32)
    DEFINE_integer("batch_size_dev", 50); DEFINE_float("learning_rate", 1e-4); DEFINE_integer("max_epoch", 500)
    DEFINE_float("entropy_s", None); DEFINE_float("entropy_sen", None); DEFINE_float("entropy_rec", None)
    # Conversation settings
    DEFINE_integer("exchange_samples", 3); DEFINE_integer("max_exchange", 3); DEFINE_boolean("fixed_exchange", True)
    DEFINE_boolean("bit_flip", False); DEFINE_string("corrupt_region", None)
    # ---- additive flags of this build ----
    DEFINE_integer("seed", 0, "seed of weight init and of the in-kernel Philox sampling")
    DEFINE_integer("max_steps", 0, "stop training after this many optimizer steps (0 = run max_epoch epochs)")
    DEFINE_string("synthetic_data", None, "write synthetic train/dev HDF5 + descriptions + GloVe files into this "
                                          "directory (if missing) and train on them")
    # data-parallel epoch loop: one process per GPU (python -m torch.distributed.run ... -m multimodalgame_amd.model ...);
    # every minibatch of -batch_size samples is sharded over the ranks (multimodalgame_amd/dist.py)
    DEFINE_integer("world_size", 0, "number of data-parallel ranks (0: WORLD_SIZE from the launcher's environment, else 1)")
    DEFINE_integer("rank", -1, "this process's rank (-1: RANK from the launcher's environment, else 0)")
    DEFINE_string("dist_backend", "nccl", "torch.distributed backend of the data-parallel job (\"nccl\" is RCCL on ROCm)")


# presets, model.py:1605-1636
def Fixed():
    FLAGS.img_feat = "avgpool_512"; FLAGS.img_feat_dim = 512; FLAGS.fixed_exchange = True; FLAGS.visual_attn = False


def Adaptive():
    FLAGS.img_feat = "avgpool_512"; FLAGS.img_feat_dim = 512; FLAGS.fixed_exchange = False; FLAGS.visual_attn = False


def FixedAttention():
    FLAGS.img_feat = "layer4_2"; FLAGS.img_feat_dim = 512; FLAGS.fixed_exchange = True; FLAGS.visual_attn = True
    FLAGS.attn_dim = 256; FLAGS.attn_extra_context = True; FLAGS.attn_context_dim = 1000


def AdaptiveAttention():
    FLAGS.img_feat = "layer4_2"; FLAGS.img_feat_dim = 512; FLAGS.fixed_exchange = False; FLAGS.visual_attn = True
    FLAGS.attn_dim = 256; FLAGS.attn_extra_context = True; FLAGS.attn_context_dim = 1000


_PRESETS = {"Fixed": Fixed, "Adaptive": Adaptive, "FixedAttention": FixedAttention, "AdaptiveAttention": AdaptiveAttention}


def default_flags(argv=None):
    """model.py:1744-1810."""
    argv = sys.argv if argv is None else argv
    if FLAGS.log_load:
        log_flags = json.loads(open(FLAGS.log_load).read())
        for k in log_flags.keys():
            if k in FLAGS.FlagValuesDict().keys():
                setattr(FLAGS, k, log_flags[k])
        FLAGS(argv)
    if FLAGS.model_type:
        _PRESETS[FLAGS.model_type]()
        FLAGS(argv)
    assert FLAGS.sender_out_dim == FLAGS.rec_w_dim, \
        "Both sender and receiver should communicate with same dim vectors for now."
    if not FLAGS.use_binary:
        FLAGS.exchange_samples = 0
    if not FLAGS.experiment_name:
        FLAGS.experiment_name = "{}-so_{}-wv_{}-bs_{}-{}".format(
            FLAGS.images, FLAGS.sender_out_dim, FLAGS.wv_dim, FLAGS.batch_size, str(int(time.time())))
    for flag, suffix in (("conf_mat", ".conf_mat.txt"), ("log_file", ".log"), ("eval_csv_file", ".eval.csv"),
                         ("json_file", ".json"), ("checkpoint", ".pt"), ("binary_output", ".bv.hdf5")):
        if not getattr(FLAGS, flag):
            setattr(FLAGS, flag, os.path.join(FLAGS.log_path, FLAGS.experiment_name + suffix))
    if not FLAGS.branch:
        FLAGS.branch = os.popen("git rev-parse --abbrev-ref HEAD 2>/dev/null").read().strip()
    if not FLAGS.sha:
        FLAGS.sha = os.popen("git rev-parse HEAD 2>/dev/null").read().strip()
    try:
        import torch
        if not torch.cuda.is_available():
            FLAGS.cuda = False
    except ImportError:
        FLAGS.cuda = False
    if FLAGS.debug:
        import numpy as np
        np.seterr(all="raise")
    FLAGS.glove_path = os.path.expanduser(FLAGS.glove_path)


# ---------------------------------------------------------------------------------------------------------------------
# Reference switches whose code paths are OUTSIDE the accelerated hot path (SURVEY.md §2 "out of scope"): the flags parse
# (so a reference command line is accepted or rejected for a stated reason), but asking for one of them raises instead of
# silently computing something else.
# ---------------------------------------------------------------------------------------------------------------------
UNSUPPORTED = (
    # (flag, predicate on its value, reference lines that read it)
    ("desc_attn", lambda v: bool(v), "model.py:344-409 (description attention)"),
    ("sender_mix", lambda v: v not in (None, "sum"), "model.py:201-214 (prod / mou mixing of h_x and h_w)"),
    ("flipout_sen", lambda v: v is not None, "model.py:233-234, 554-568 (random bit flips of the sender message)"),
    ("flipout_rec", lambda v: v is not None, "model.py:467-470, 554-568 (random bit flips of the receiver message)"),
    ("ignore_receiver", lambda v: bool(v), "model.py:217-218 (sender ignores the receiver's message)"),
    ("ignore_code", lambda v: bool(v), "model.py:219-221 (sender ignores its code input)"),
    ("visual_attn", lambda v: bool(v), "model.py:114-191 (visual attention over layer4_2)"),
    ("bit_flip", lambda v: bool(v), "model.py:813-824 (message corruption)"),
)


def check_supported(flags=None):
    """Raise NotImplementedError naming every requested switch the MI355X path does not implement."""
    fl = FLAGS if flags is None else flags
    bad = []
    for name, pred, where in UNSUPPORTED:
        try:
            v = getattr(fl, name)
        except (AttributeError, KeyError, FlagsError):
            continue
        if pred(v):
            bad.append("-%s=%s [reference: %s]" % (name, v, where))
    if bad:
        raise NotImplementedError("outside the accelerated exchange path (SURVEY.md §2): " + "; ".join(bad))
