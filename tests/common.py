"""Shared helpers for the parity tests: load a golden case, re-run it with the
CPU oracle (or any engine with the same ``train_minibatch`` contract) and compare
packed results entry by entry."""
import json
import os

import numpy as np
import torch

from oracle import cpu_ref

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

U_OVERRIDES = {}


def _one_active(i, u_z, u_s, u_w):
    u_s = u_s.copy()
    u_s[0, :, 0] = 0.999999
    u_s[0, 2, 0] = 0.0
    u_s[1, :, 0] = 0.0
    u_s[2, :, 0] = 0.999999
    return u_z, u_s, u_w


def _all_stop(i, u_z, u_s, u_w):
    u_s = u_s.copy()
    u_s[:, :, 0] = 0.999999
    return u_z, u_s, u_w


def _never_stop(i, u_z, u_s, u_w):
    u_s = u_s.copy()
    u_s[:, :, 0] = 0.0
    return u_z, u_s, u_w


U_OVERRIDES = {"g5_one_active": _one_active, "g5_all_stop_first": _all_stop, "g5_never_stop": _never_stop}

TRAIN_CASES = ["g2_adaptive_c1", "g3_fixed_c3shard", "g3_continuous", "g3_tiny_sgd", "g3_tiny_adam",
               "g5_one_active", "g5_all_stop_first", "g5_never_stop"]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


FLAG_KEYS = set(cpu_ref.Flags().__dict__.keys())


def flags_from_meta(meta):
    return cpu_ref.Flags(**{k: v for k, v in meta.items() if k in FLAG_KEYS})


def case_inputs(meta, i, name=None):
    """Inputs of minibatch ``i`` of a golden train case: x, target, desc, (u_z,u_s,u_w)."""
    fl = flags_from_meta(meta)
    x, target, desc = cpu_ref.synthetic_batch(meta["batch"], meta["n_classes"], fl.img_feat_dim, fl.wv_dim,
                                              seed=meta["seed_data"] + i)
    u = cpu_ref.draw_uniforms(fl.max_exchange, meta["batch"], fl.rec_w_dim, seed=meta["seed_uniforms"] + i)
    if name in U_OVERRIDES:
        u = U_OVERRIDES[name](i, *u)
    return x, target, desc, u


def oracle_train_case(name, meta):
    """Re-run a golden train case with the CPU oracle; returns the packed dict."""
    fl = flags_from_meta(meta)
    torch.manual_seed(0)
    tape = cpu_ref.UniformTape()
    models = cpu_ref.build_agents(fl, rng=tape)
    cpu_ref.load_filled(models, seed=meta["seed_weights"])
    optimizers = cpu_ref.build_optimizers(models, fl)
    out = {}
    for i in range(meta["n_minibatches"]):
        x, target, desc, (u_z, u_s, u_w) = case_inputs(meta, i, name)
        tape.u = {"z": u_z, "s": u_s, "w": u_w}
        tape.t = {"z": 0, "s": 0, "w": 0}
        res = cpu_ref.train_minibatch(models, optimizers, torch.from_numpy(x), torch.from_numpy(target),
                                      torch.from_numpy(desc), fl)
        out.update(cpu_ref.pack_train(res, models, prefix="mb%d." % i))
    return out


def compare_packed(got, want, atol=1e-5, rtol=1e-4, skip=(), only_prefix=None):
    """Compare two packed dicts.  Bit/mask/count entries must match exactly; float entries
    within atol + rtol*|want|."""
    exact = ("s_masks", "s_feats", "sen_feats", "n_steps", "hits")
    problems = []
    for k in want.keys() if hasattr(want, "keys") else want.files:
        if k == "meta" or k.endswith(".u_s"):
            continue
        if only_prefix and not k.startswith(only_prefix):
            continue
        if any(s in k for s in skip):
            continue
        if k not in got:
            problems.append("missing " + k)
            continue
        a, b = np.asarray(got[k]), np.asarray(want[k])
        if a.shape != b.shape:
            problems.append("%s shape %s vs %s" % (k, a.shape, b.shape))
            continue
        tail = k.split(".")[-1]
        if tail in exact:
            if not np.array_equal(a, b):
                problems.append("%s differs (exact) in %d places" % (k, int((a != b).sum())))
        else:
            err = np.abs(a.astype(np.float64) - b.astype(np.float64))
            tol = atol + rtol * np.abs(b.astype(np.float64))
            if a.size and not np.all(err <= tol):
                problems.append("%s max err %.3e (tol %.1e)" % (k, float(err.max()), float(tol.flat[err.argmax()])))
    return problems
