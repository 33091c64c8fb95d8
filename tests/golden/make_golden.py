#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own code.

Run in the build container only (needs /root/reference; the GPU box never runs
this).  Nothing from the reference is copied into the repo: ``model.py`` is read
as text, 15 semantics-restoring substitutions are applied
in memory so the Python-2.7 / PyTorch-0.1.12 code executes on Python 3.10 /
torch 2.10, and only NUMBERS (inputs' seeds, outputs) are written out.

The harness drives the reference's ``Sender/Receiver/Baseline/exchange/
get_rec_outp/multistep_loss_*`` exactly as ``run()`` does at model.py:1219-1330
(``run()`` itself is Python-2-only).  Weights, inputs and sampling uniforms come
from the repo-owned deterministic fillers in ``oracle/cpu_ref.py`` so that tests
can regenerate them from seeds.

usage: python tests/golden/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import cpu_ref  # noqa: E402  (fillers only; the oracle is NOT what generates outputs)

warnings.filterwarnings("ignore")


# ----------------------------------------------------------------------------
# Loading the reference
# ----------------------------------------------------------------------------
class _FlagValues(object):
    def __init__(self):
        object.__setattr__(self, "_d", {})

    def __getattr__(self, k):
        try:
            return self._d[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._d[k] = v

    def FlagValuesDict(self):
        return dict(self._d)

    def __call__(self, argv):
        return argv


def _stub_modules():
    gflags = types.ModuleType("gflags")
    gflags.FLAGS = _FlagValues()

    def _define(name, default, *a, **k):
        setattr(gflags.FLAGS, name, default)
    for fn in ("DEFINE_string", "DEFINE_boolean", "DEFINE_integer", "DEFINE_float"):
        setattr(gflags, fn, _define)
    gflags.DEFINE_enum = lambda name, default, choices, help="": setattr(gflags.FLAGS, name, default)
    sys.modules["gflags"] = gflags
    sys.modules["h5py"] = types.ModuleType("h5py")
    nltk = types.ModuleType("nltk")
    tok = types.ModuleType("nltk.tokenize"); tok.word_tokenize = lambda s: s.split()
    corp = types.ModuleType("nltk.corpus"); corp.stopwords = types.SimpleNamespace(words=lambda lang: [])
    sys.modules.update({"nltk": nltk, "nltk.tokenize": tok, "nltk.corpus": corp})
    tv = types.ModuleType("torchvision")
    for sub in ("models", "datasets", "transforms"):
        m = types.ModuleType("torchvision." + sub)
        setattr(tv, sub, m)
        sys.modules["torchvision." + sub] = m
    sys.modules["torchvision"] = tv
    return gflags


class _RandomProxy(object):
    """Replaces ``np.random`` inside the reference module: ``rand`` pops from a queue of
    pre-drawn uniforms (call order z, s, w per step -- model.py:227, 420, 460)."""

    def __init__(self):
        self.queue = []

    def rand(self, *shape):
        arr = self.queue.pop(0)
        assert tuple(arr.shape) == tuple(shape), (arr.shape, shape)
        return arr.astype(np.float64)

    def __getattr__(self, k):
        return getattr(np.random, k)


class _NpProxy(object):
    def __init__(self):
        self.random = _RandomProxy()

    def __getattr__(self, k):
        return getattr(np, k)


def _sub(src, old, new, count=1):
    assert src.count(old) == count, "pattern occurs %d times (want %d): %r" % (src.count(old), count, old)
    return src.replace(old, new)


def load_reference(ref_dir):
    gflags = _stub_modules()
    sys.path.insert(0, ref_dir)
    import misc  # the reference's misc.py (imports fine with the stubs)

    def xavier_normal(tensor, gain=1):   # misc.py:379-381 recurses forever on torch>=0.4
        fan_in, fan_out = misc._calculate_fan_in_and_fan_out(tensor)
        std = gain * np.sqrt(2.0 / (fan_in + fan_out))
        return tensor.normal_(0, std)
    misc.xavier_normal = xavier_normal

    src = open(os.path.join(ref_dir, "model.py")).read()
    src = src[:src.index("if __name__ == '__main__':")]
    # --- SURVEY.md Appendix B2 substitutions -------------------------------
    src = _sub(src, "receiver.h_z if receiver.h_z else", "receiver.h_z if receiver.h_z is not None else")
    src = _sub(src, "if not self.s_prob_prod or not FLAGS.s_prob_prod:",
               "if self.s_prob_prod is None or not FLAGS.s_prob_prod:")
    src = _sub(src, "stop_mask[-1].float().sum().data[0] == 0", "stop_mask[-1].float().sum().item() == 0")
    src = _sub(src, "negentropy = map(negent, y)", "negentropy = list(map(negent, y))")
    src = _sub(src, "torch.masked_select(inp, mask.detach())", "torch.masked_select(inp, mask.detach().bool())")
    src = _sub(src, "log_p_z = log_p_z.sum(1)", "log_p_z = log_p_z.sum(1, keepdim=True)")
    src = _sub(src, "weight / np.maximum(1., torch.std(weight.data))",
               "weight / max(1., torch.std(weight.data).item())")
    src = _sub(src, "_mask_sums = [m.float().sum().data[0] for m in masks]",
               "_mask_sums = [m.float().sum().item() for m in masks]")
    src = _sub(src, "            feat = feat[mask.expand_as(feat)]",
               "            mask = mask.bool()\n            feat = feat[mask.expand_as(feat)]")
    src = _sub(src, "outp = map(mapped_fn, binary_features, binary_probs,\n                   baseline_scores, masks, _mask_sums)",
               "outp = list(map(mapped_fn, binary_features, binary_probs,\n                   baseline_scores, masks, _mask_sums))")
    src = _sub(src, "outp = map(lambda feat, prob, scores: calculate_loss_binary(feat, prob, logs, scores, entropy_penalty),\n                   binary_features, binary_probs, baseline_scores)",
               "outp = list(map(lambda feat, prob, scores: calculate_loss_binary(feat, prob, logs, scores, entropy_penalty),\n                   binary_features, binary_probs, baseline_scores))")
    src = _sub(src, "losses = map(lambda scores, mask: calculate_loss_bas(\n            scores[mask].view(-1, 1), logs[mask].view(-1, 1)),\n            baseline_scores, masks)",
               "losses = list(map(lambda scores, mask: calculate_loss_bas(\n            scores[mask.bool()].view(-1, 1), logs[mask.bool()].view(-1, 1)),\n            baseline_scores, masks))")
    src = _sub(src, "losses = map(lambda scores: calculate_loss_bas(scores, logs),\n                     baseline_scores)",
               "losses = list(map(lambda scores: calculate_loss_bas(scores, logs),\n                     baseline_scores))")
    mod = types.ModuleType("reference_model")
    mod.__file__ = os.path.join(ref_dir, "model.py")
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    mod.np = _NpProxy()
    mod.flags()     # the reference's own flag definitions -> defaults land in FLAGS
    return mod, gflags.FLAGS


# ----------------------------------------------------------------------------
# Harness: model.py:1014-1064 (construction) and 1219-1339 (one minibatch)
# ----------------------------------------------------------------------------
def set_flags(FLAGS, fl):
    FLAGS.cuda = False
    FLAGS.debug = False
    for k in ("use_binary", "fixed_exchange", "max_exchange", "first_rec", "s_prob_prod",
              "entropy_s", "entropy_sen", "entropy_rec", "optim_type", "learning_rate", "batch_size",
              "top_k_train", "top_k_dev", "img_feat_dim", "img_h_dim", "rec_w_dim", "sender_out_dim",
              "rec_hidden", "rec_out_dim", "rec_s_dim", "wv_dim", "baseline_hid_dim", "ignore_receiver",
              "flipout_sen", "flipout_rec", "flipout_dev"):
        setattr(FLAGS, k, getattr(fl, k))
    FLAGS.img_feat = "avgpool_512"
    FLAGS.visual_attn = False
    FLAGS.desc_attn = False
    FLAGS.sender_mix = "sum"
    FLAGS.ignore_code = False
    FLAGS.attn_extra_context = False
    FLAGS.bit_flip = False


def build_ref_models(ref, FLAGS):
    sender = ref.Sender(feature_type=FLAGS.img_feat, feat_dim=FLAGS.img_feat_dim, h_dim=FLAGS.img_h_dim,
                        w_dim=FLAGS.rec_w_dim, bin_dim_out=FLAGS.sender_out_dim, use_binary=FLAGS.use_binary,
                        use_attn=False, attn_dim=256, attn_extra_context=False, attn_context_dim=4096)
    baseline_sen = ref.Baseline(hid_dim=FLAGS.baseline_hid_dim, x_dim=FLAGS.img_h_dim,
                                binary_dim=FLAGS.rec_w_dim, inp_dim=0)
    receiver = ref.Receiver(hid_dim=FLAGS.rec_hidden, out_dim=FLAGS.rec_out_dim, z_dim=FLAGS.sender_out_dim,
                            desc_dim=FLAGS.wv_dim, w_dim=FLAGS.rec_w_dim, s_dim=FLAGS.rec_s_dim,
                            use_binary=FLAGS.use_binary)
    baseline_rec = ref.Baseline(hid_dim=FLAGS.baseline_hid_dim, x_dim=0,
                                binary_dim=FLAGS.rec_w_dim, inp_dim=FLAGS.rec_hidden)
    return dict(sender=sender, receiver=receiver, baseline_sen=baseline_sen, baseline_rec=baseline_rec)


def queue_uniforms(ref, fl, u_z, u_s, u_w):
    q = []
    for t in range(u_s.shape[0]):
        if fl.use_binary:
            q += [u_z[t], u_s[t], u_w[t]]
        else:
            q += [u_s[t]]
    ref.np.random.queue = q


def ref_train_minibatch(ref, FLAGS, models, optimizers, data, target, desc):
    sender, receiver = models["sender"], models["receiver"]
    baseline_sen, baseline_rec = models["baseline_sen"], models["baseline_rec"]
    exchange_args = dict(data=data, target=target, desc=desc, desc_set=None, desc_set_lens=None,
                         train=True, break_early=not FLAGS.fixed_exchange)
    s, sen_w, rec_w, y, bs, br = ref.exchange(sender, receiver, baseline_sen, baseline_rec, exchange_args)
    s_masks, s_feats, s_probs = s
    sen_feats, sen_probs = sen_w
    rec_feats, rec_probs = rec_w
    if FLAGS.fixed_exchange:
        binary_s_masks = binary_rec_masks = binary_sen_masks = bas_rec_masks = bas_sen_masks = y_masks = None
    else:
        binary_s_masks = s_masks[:-1]
        binary_rec_masks = s_masks[1:-1]
        binary_sen_masks = s_masks[:-1]
        bas_rec_masks = s_masks[:-1]
        bas_sen_masks = s_masks[:-1]
        y_masks = [torch.min(1 - m1, m2) for m1, m2 in zip(s_masks[1:], s_masks[:-1])]
    outp, ent_y_rec = ref.get_rec_outp(y, y_masks)
    dist = F.log_softmax(outp, dim=1)
    nll_loss = nn.NLLLoss()(dist, target)
    logs = ref.loglikelihood(dist.detach(), target.view(-1, 1))
    zero = torch.zeros(1)
    loss_binary_s = loss_binary_rec = loss_binary_sen = loss_bas_rec = loss_bas_sen = zero
    if FLAGS.use_binary:
        if not FLAGS.fixed_exchange:
            loss_binary_s, _ = ref.multistep_loss_binary(s_feats, s_probs, logs, br, binary_s_masks, FLAGS.entropy_s)
        if len(rec_feats[:-1]) > 0:
            loss_binary_rec, _ = ref.multistep_loss_binary(
                rec_feats[:-1], rec_probs[:-1], logs, br[:-1], binary_rec_masks, FLAGS.entropy_rec)
        else:
            loss_binary_rec = torch.zeros(1)
        loss_binary_sen, _ = ref.multistep_loss_binary(sen_feats, sen_probs, logs, bs, binary_sen_masks, FLAGS.entropy_sen)
        loss_bas_rec = ref.multistep_loss_bas(br, logs, bas_rec_masks)
        loss_bas_sen = ref.multistep_loss_bas(bs, logs, bas_sen_masks)
    loss_rec = nll_loss
    if FLAGS.use_binary:
        loss_rec = loss_rec + loss_binary_rec
        if not FLAGS.fixed_exchange:
            loss_rec = loss_rec + loss_binary_s
        loss_sen = loss_binary_sen
    else:
        loss_sen = zero

    grads, grad_norms = {}, {}

    def _update(opt_key, model_key, loss):
        opt, model = optimizers[opt_key], models[model_key]
        opt.zero_grad()
        loss.backward()
        grads[model_key] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        grad_norms[model_key] = float(nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.))
        opt.step()
    _update("optimizer_rec", "receiver", loss_rec)
    if FLAGS.use_binary:
        _update("optimizer_sen", "sender", loss_sen)
        _update("optimizer_bas_rec", "baseline_rec", loss_bas_rec)
        _update("optimizer_bas_sen", "baseline_sen", loss_bas_sen)
    top_k_ind = dist.detach().numpy().argsort()[:, -FLAGS.top_k_train:]
    hits = int((top_k_ind == target.view(-1, 1).numpy()).sum())
    return dict(n_steps=len(y), s_masks=s_masks, s_feats=s_feats, s_probs=s_probs, sen_feats=sen_feats,
                sen_probs=sen_probs, rec_feats=rec_feats, rec_probs=rec_probs, y=y, bs=bs, br=br,
                outp=outp, dist=dist, logs=logs, nll_loss=nll_loss, loss_binary_s=loss_binary_s,
                loss_binary_rec=loss_binary_rec, loss_binary_sen=loss_binary_sen,
                loss_bas_rec=loss_bas_rec, loss_bas_sen=loss_bas_sen, grads=grads, grad_norms=grad_norms,
                hits=hits)


_stack = cpu_ref._stack
pack_train = cpu_ref.pack_train


def make_flags(**kw):
    return cpu_ref.Flags(**kw)


def run_train_case(ref, FLAGS, fl, n_classes, batch, seeds, n_minibatches=1, u_override=None):
    """seeds = dict(weights=, data=, uniforms=).  Returns npz dict."""
    set_flags(FLAGS, fl)
    torch.manual_seed(0)
    models = build_ref_models(ref, FLAGS)
    cpu_ref.load_filled(models, seed=seeds["weights"])
    cls = {"SGD": optim.SGD, "Adam": optim.Adam, "RMSprop": optim.RMSprop}[fl.optim_type]
    optimizers = dict(optimizer_rec=cls(models["receiver"].parameters(), lr=fl.learning_rate),
                      optimizer_sen=cls(models["sender"].parameters(), lr=fl.learning_rate),
                      optimizer_bas_rec=cls(models["baseline_rec"].parameters(), lr=fl.learning_rate),
                      optimizer_bas_sen=cls(models["baseline_sen"].parameters(), lr=fl.learning_rate))
    out = {}
    for i in range(n_minibatches):
        x, target, desc = cpu_ref.synthetic_batch(batch, n_classes, fl.img_feat_dim, fl.wv_dim, seed=seeds["data"] + i)
        u_z, u_s, u_w = cpu_ref.draw_uniforms(fl.max_exchange, batch, fl.rec_w_dim, seed=seeds["uniforms"] + i)
        if u_override is not None:
            u_z, u_s, u_w = u_override(i, u_z, u_s, u_w)
            out["mb%d.u_s" % i] = u_s
        queue_uniforms(ref, fl, u_z, u_s, u_w)
        res = ref_train_minibatch(ref, FLAGS, models, optimizers,
                                  torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(desc))
        out.update(pack_train(res, models, prefix="mb%d." % i))
    return out


def flags_to_meta(fl, n_classes, batch, seeds, n_minibatches):
    meta = {k: v for k, v in fl.__dict__.items()}
    meta.update(n_classes=n_classes, batch=batch, n_minibatches=n_minibatches,
                seed_weights=seeds["weights"], seed_data=seeds["data"], seed_uniforms=seeds["uniforms"])
    import json
    return np.array(json.dumps(meta))


# ----------------------------------------------------------------------------
# Cases
# ----------------------------------------------------------------------------
TINY = dict(img_feat_dim=16, img_h_dim=8, rec_w_dim=6, sender_out_dim=6, rec_hidden=5, wv_dim=7,
            baseline_hid_dim=9)
C1 = dict(img_feat_dim=512, img_h_dim=256, rec_w_dim=32, sender_out_dim=32, rec_hidden=64, wv_dim=100,
          baseline_hid_dim=500, batch_size=64, max_exchange=10, learning_rate=1e-4,
          entropy_rec=0.01, entropy_sen=0.01, entropy_s=0.08, top_k_train=6, top_k_dev=6)


def case_g1_agents(ref, FLAGS):
    """G1: agent forwards at tiny dims, explicit weights stored."""
    fl = make_flags(use_binary=True, fixed_exchange=False, max_exchange=3, batch_size=4, **TINY)
    set_flags(FLAGS, fl)
    models = build_ref_models(ref, FLAGS)
    filled = cpu_ref.load_filled(models, seed=7)
    x, target, desc = cpu_ref.synthetic_batch(4, 3, fl.img_feat_dim, fl.wv_dim, seed=5)
    u_z, u_s, u_w = cpu_ref.draw_uniforms(3, 4, fl.rec_w_dim, seed=9)
    xt, dt = torch.from_numpy(x), torch.from_numpy(desc)
    out = {"x": x, "desc": desc, "u_z": u_z, "u_s": u_s, "u_w": u_w}
    for a, d in filled.items():
        for k, v in d.items():
            out["w.%s.%s" % (a, k)] = v
    s, r = models["sender"], models["receiver"]
    # Sender, train: t=0 then t=1 with a given w
    s.train(); s.reset_state()
    ref.np.random.queue = [u_z[0], u_z[1]]
    z0, p0 = s(xt, torch.zeros(4, 6), None, 0)
    out["sen.train.t0.z"], out["sen.train.t0.p"] = z0.numpy(), p0.detach().numpy()
    w_in = torch.from_numpy((u_w[0] < 0.5).astype(np.float32))
    out["sen.w_in"] = w_in.numpy()
    z1, p1 = s(xt, w_in, None, 1)
    out["sen.train.t1.z"], out["sen.train.t1.p"] = z1.numpy(), p1.detach().numpy()
    out["sen.h_x"] = s.h_x.detach().numpy()
    s.eval()
    ze, pe = s(xt, w_in, None, 1)
    out["sen.eval.t1.z"], out["sen.eval.t1.p"] = ze.numpy(), pe.detach().numpy()
    # Receiver, train: two consecutive steps
    for mode in ("train", "eval"):
        r.train() if mode == "train" else r.eval()
        r.reset_state()
        ref.np.random.queue = [u_s[0], u_w[0], u_s[1], u_w[1]]
        zin = [z0, z1]
        for t in range(2):
            (sb, sp), (wf, wp), y = r(zin[t], dt, None, None)
            pre = "rec.%s.t%d." % (mode, t)
            out[pre + "s"], out[pre + "s_prob"] = sb.numpy(), sp.detach().numpy()
            out[pre + "w"], out[pre + "w_prob"] = wf.detach().numpy(), wp.detach().numpy()
            out[pre + "y"] = y.detach().numpy()
            out[pre + "h_z"] = r.h_z.detach().numpy()
            out[pre + "h_w"] = r.h_w.detach().numpy()
    out["bas_sen"] = models["baseline_sen"](s.h_x.detach(), w_in, None).detach().numpy()
    out["bas_rec"] = models["baseline_rec"](None, z1, r.h_z.detach()).detach().numpy()
    out["meta"] = flags_to_meta(fl, 3, 4, dict(weights=7, data=5, uniforms=9), 0)
    return out


def case_g4_eval(ref, FLAGS):
    """G4: eval pass (round, cumulative-product stop bit) at C1 shape, dev batch 50."""
    fl = make_flags(use_binary=True, fixed_exchange=False, **C1)
    set_flags(FLAGS, fl)
    models = build_ref_models(ref, FLAGS)
    cpu_ref.load_filled(models, seed=3)
    # bias the stop head so that the product of stop probabilities crosses 0.5 at different steps
    with torch.no_grad():
        models["receiver"].s.bias.fill_(1.2)
    out = {}
    x, target, desc = cpu_ref.synthetic_batch(50, 30, 512, 100, seed=77)
    exchange_args = dict(data=torch.from_numpy(x), target=torch.from_numpy(target), desc=torch.from_numpy(desc),
                         desc_set=None, desc_set_lens=None, train=False, break_early=True,
                         corrupt=False, corrupt_region=None)
    with torch.no_grad():
        s, sen_w, rec_w, y, _, _ = ref.exchange(models["sender"], models["receiver"], None, None, exchange_args)
        s_masks, s_feats, s_probs = s
        y_masks = [torch.min(1 - m1, m2) for m1, m2 in zip(s_masks[1:], s_masks[:-1])]
        outp, _ = ref.get_rec_outp(y, y_masks)
        dist = F.log_softmax(outp, dim=1)
    top_k_ind = dist.numpy().argsort()[:, -6:]
    out["n_steps"] = np.int64(len(y))
    out["s_masks"] = _stack(s_masks).astype(np.uint8)
    out["s_feats"], out["s_probs"] = _stack(s_feats), _stack(s_probs)
    out["sen_feats"], out["sen_probs"] = _stack(sen_w[0]), _stack(sen_w[1])
    out["rec_feats"], out["rec_probs"] = _stack(rec_w[0]), _stack(rec_w[1])
    out["y"] = _stack(y)
    out["outp"], out["dist"] = outp.numpy(), dist.numpy()
    out["top_k_ind"] = top_k_ind.astype(np.int64)
    out["hits"] = np.int64((top_k_ind == target.reshape(-1, 1)).sum())
    out["conversation_lengths"] = torch.cat(s_feats, 1).float().sum(1).numpy()
    out["meta"] = flags_to_meta(fl, 30, 50, dict(weights=3, data=77, uniforms=0), 0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    ref, FLAGS = load_reference(args.ref)

    def save(name, d):
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **d)
        print("%-28s %7.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024.0, len(d)))

    save("g1_agents_tiny", case_g1_agents(ref, FLAGS))

    # G2: full Adaptive train step at config-1 shape (README.md:30-53 command line)
    fl = make_flags(use_binary=True, fixed_exchange=False, **C1)
    seeds = dict(weights=0, data=1234, uniforms=0)
    d = run_train_case(ref, FLAGS, fl, 30, 64, seeds, n_minibatches=2)
    d["meta"] = flags_to_meta(fl, 30, 64, seeds, 2)
    save("g2_adaptive_c1", d)

    # G3a: Fixed exchange, one 64-sample shard of config 3
    fl = make_flags(use_binary=True, fixed_exchange=True, **C1)
    seeds = dict(weights=1, data=4321, uniforms=11)
    d = run_train_case(ref, FLAGS, fl, 30, 64, seeds, n_minibatches=1)
    d["meta"] = flags_to_meta(fl, 30, 64, seeds, 1)
    save("g3_fixed_c3shard", d)

    # G3b: continuous messages (-nouse_binary), Fixed, many classes (config-5 flavour, reduced)
    c5 = dict(C1); c5.update(batch_size=32, max_exchange=4, entropy_rec=None, entropy_sen=None, entropy_s=None)
    fl = make_flags(use_binary=False, fixed_exchange=True, **c5)
    seeds = dict(weights=2, data=99, uniforms=5)
    d = run_train_case(ref, FLAGS, fl, 200, 32, seeds, n_minibatches=2)
    d["meta"] = flags_to_meta(fl, 200, 32, seeds, 2)
    save("g3_continuous", d)

    # G3c: Adaptive without entropy penalties, SGD and Adam variants at tiny dims
    for opt_name in ("SGD", "Adam"):
        fl = make_flags(use_binary=True, fixed_exchange=False, max_exchange=5, batch_size=8,
                        optim_type=opt_name, learning_rate=1e-2, top_k_train=2, **TINY)
        seeds = dict(weights=4, data=8, uniforms=15)
        d = run_train_case(ref, FLAGS, fl, 5, 8, seeds, n_minibatches=3)
        d["meta"] = flags_to_meta(fl, 5, 8, seeds, 3)
        save("g3_tiny_" + opt_name.lower(), d)

    save("g4_eval_c1", case_g4_eval(ref, FLAGS))

    # G5a: a step with exactly one active sample (std guard, model.py:914)
    def one_active(i, u_z, u_s, u_w):
        u_s = u_s.copy()
        u_s[0, :, 0] = 0.999999   # everybody stops at step 0 ...
        u_s[0, 2, 0] = 0.0        # ... except sample 2
        u_s[1, :, 0] = 0.0        # sample 2 continues through step 1
        u_s[2, :, 0] = 0.999999   # and stops at step 2
        return u_z, u_s, u_w
    fl = make_flags(use_binary=True, fixed_exchange=False, max_exchange=6, batch_size=6, entropy_s=0.08,
                    entropy_rec=0.01, entropy_sen=0.01, top_k_train=2, **TINY)
    seeds = dict(weights=21, data=22, uniforms=23)
    d = run_train_case(ref, FLAGS, fl, 4, 6, seeds, n_minibatches=1, u_override=one_active)
    d["meta"] = flags_to_meta(fl, 4, 6, seeds, 1)
    save("g5_one_active", d)

    # G5b: every sample stops after the first step (model.py:1284-1289: no receiver-message loss)
    def all_stop(i, u_z, u_s, u_w):
        u_s = u_s.copy()
        u_s[:, :, 0] = 0.999999
        return u_z, u_s, u_w
    seeds = dict(weights=31, data=32, uniforms=33)
    d = run_train_case(ref, FLAGS, fl, 4, 6, seeds, n_minibatches=1, u_override=all_stop)
    d["meta"] = flags_to_meta(fl, 4, 6, seeds, 1)
    save("g5_all_stop_first", d)

    # G5c: nobody ever stops (forced final mask, model.py:870)
    def never_stop(i, u_z, u_s, u_w):
        u_s = u_s.copy()
        u_s[:, :, 0] = 0.0
        return u_z, u_s, u_w
    seeds = dict(weights=41, data=42, uniforms=43)
    d = run_train_case(ref, FLAGS, fl, 4, 6, seeds, n_minibatches=1, u_override=never_stop)
    d["meta"] = flags_to_meta(fl, 4, 6, seeds, 1)
    save("g5_never_stop", d)


if __name__ == "__main__":
    main()
