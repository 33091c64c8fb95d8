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
    if name is not None and name in U_OVERRIDES:
        u = U_OVERRIDES[name](i, *u)
    return x, target, desc, u


# |pre-activation| below which two correct implementations may put a ReLU unit on different sides: the forward tolerance itself.
# A pre-activation is a forward quantity (a sum of ~10^2 products of parameters that, from the second minibatch on, agree to
# ~1e-5 only -- RMSprop normalises rounding-noise gradients into O(lr) steps), so two implementations within 1e-4 of each other
# can disagree on the sign of any unit with |pre| < 1e-4.  The window only selects CANDIDATES: a unit is touched only if the
# GPU's own pre-activation (captured from its tape) has the other sign, and then the oracle re-run with that mask must agree
# on every entry (assert_parity) -- widening the window excuses nothing.
RELU_EPS = 1e-4


def _relu_flips(res, y1_out, bas_out, n_classes, binary=True, fixed=False):
    """Near-threshold ReLU units of one oracle minibatch, from the outputs its own y1 / baseline linear1 layers produced
    (forward hooks): the units whose mask may differ in another correct fp32 implementation.
      "y":        {r}  columns of the y head with |W_y1 [h_t* || desc_d] + b|[b, d, r] < RELU_EPS at the sample's OUTPUT step
      "bas_rec" / "bas_sen": {k}  hidden units of a baseline with |pre| < RELU_EPS on a live (step, sample) row
      "pos":      the same units with their position and the side the ORACLE put them on:
                  ("y", t*, b, d, r, pre > 0) / ("bas_rec" | "bas_sen", t, b, k, pre > 0)."""
    masks = np.stack([m.detach().numpy().reshape(-1) for m in res["s_masks"]])      # [n + 1, B]; the last one is forced to 0
    n, B = res["n_steps"], masks.shape[1]
    stopped = masks[1:n + 1] == 0
    tstar = np.where(stopped.any(0), stopped.argmax(0), n - 1)
    if fixed:                                   # Fixed exchange: the output is the LAST step's, every row is live (model.py:903, 963-967)
        tstar = np.full(B, n - 1)
    flips = {"y": set(), "bas_rec": set(), "bas_sen": set(), "where": [], "pos": []}
    for b in range(B):
        raw = y1_out[int(tstar[b])].view(B, n_classes, -1)[b]                       # [D, R] at the output step (build_inp rows b * D + d)
        pre = raw.abs()
        d_idx, r_idx = torch.nonzero(pre < RELU_EPS, as_tuple=True)
        for d, r in zip(d_idx.tolist(), r_idx.tolist()):
            flips["y"].add(r); flips["where"].append("y1 unit r=%d at sample %d class %d: |pre| = %.2e" % (r, b, d, float(pre[d, r])))
            flips["pos"].append(("y", int(tstar[b]), b, d, r, bool(raw[d, r] > 0)))
    for which in (("bas_rec", "bas_sen") if binary else ()):
        for t, out in enumerate(bas_out[which][:n]):
            live = torch.from_numpy(np.asarray(t <= tstar))
            small = (out.abs() < RELU_EPS) & live.view(-1, 1)
            for b, k in torch.nonzero(small).tolist():
                flips[which].add(k); flips["where"].append("%s hidden unit k=%d at step %d sample %d: |pre| = %.2e" % (which, k, t, b, float(out[b, k].abs())))
                flips["pos"].append((which, t, b, k, bool(out[b, k] > 0)))
    return flips


_FORCE_TINY = 1e-30       # a forced unit's pre-activation: +tiny (mask on, value ~0) / -tiny (mask off)


def oracle_train_case(name, meta, flips=None, force=None, params_before=None):
    """Re-run a golden train case with the CPU oracle; returns the packed dict.
    flips: a list that receives, per minibatch, the near-threshold ReLU units (_relu_flips) and the case they belong to.
    force: per minibatch, a set of positions ("y", t, b, d, r, side) / ("bas_*", t, b, k, side) whose ReLU mask is forced to
    `side` (forced_masks: the side ANOTHER fp32 implementation put a near-threshold unit on).  The unit's pre-activation is
    replaced by +-1e-30 with the gradient path kept, so the forward pass moves by at most RELU_EPS * |w2| per unit and the
    backward pass sees the other implementation's mask -- the re-run must then agree within the normal tolerance
    (assert_parity), nothing is excused blanket-wise."""
    fl = flags_from_meta(meta)
    torch.manual_seed(0)
    tape = cpu_ref.UniformTape()
    models = cpu_ref.build_agents(fl, rng=tape)
    cpu_ref.load_filled(models, seed=meta["seed_weights"])
    optimizers = cpu_ref.build_optimizers(models, fl)
    out = {}
    y1_out, bas_out = [], {"bas_rec": [], "bas_sen": []}
    hooks = []
    state = {"mb": 0}
    D = meta["n_classes"]

    def hook(kind, store):
        def fn(mod, inp, o):
            t = len(store)
            todo = [p for p in (force[state["mb"]] if force else ()) if p[0] == kind and p[1] == t]
            if todo:
                delta = torch.zeros_like(o)
                for p in todo:
                    row, col = (p[2] * D + p[3], p[4]) if kind == "y" else (p[2], p[3])
                    delta[row, col] = (_FORCE_TINY if p[-1] else -_FORCE_TINY) - float(o[row, col].detach())
                o = o + delta                                   # gradient path kept (d/d pre = 1 where the mask is on)
            store.append(o.detach().clone())
            return o
        return fn
    if flips is not None or force:
        hooks.append(models["receiver"].y1.register_forward_hook(hook("y", y1_out)))
        hooks.append(models["baseline_rec"].linear1.register_forward_hook(hook("bas_rec", bas_out["bas_rec"])))
        hooks.append(models["baseline_sen"].linear1.register_forward_hook(hook("bas_sen", bas_out["bas_sen"])))
    for i in range(meta["n_minibatches"]):
        x, target, desc, (u_z, u_s, u_w) = case_inputs(meta, i, name)
        tape.u = {"z": u_z, "s": u_s, "w": u_w}
        tape.t = {"z": 0, "s": 0, "w": 0}
        state["mb"] = i
        if params_before is not None:                               # (oracle_losses_f64: the parameters minibatch i starts from)
            params_before.append({a: {k: v.detach().clone() for k, v in m.state_dict().items()} for a, m in models.items()})
        del y1_out[:], bas_out["bas_rec"][:], bas_out["bas_sen"][:]
        res = cpu_ref.train_minibatch(models, optimizers, torch.from_numpy(x), torch.from_numpy(target),
                                      torch.from_numpy(desc), fl)
        if flips is not None:
            f = _relu_flips(res, y1_out, bas_out, meta["n_classes"], binary=bool(fl.use_binary), fixed=bool(fl.fixed_exchange))
            f["case"] = (name, meta)
            flips.append(f)
        out.update(cpu_ref.pack_train(res, models, prefix="mb%d." % i))
    for h in hooks:
        h.remove()
    return out


def oracle_losses_f64(name, meta, want, params_oracle, params_gpu):
    """The float64 gate of the six losses (shapes whose losses are far above fp32's absolute resolution: config 4's means of
    256-bit log-likelihood sums times a reward weight, |loss| ~ 600, one fp32 ulp = 6e-5).  For every minibatch i the oracle's
    loss computation is re-run in FLOAT64 on the DISCRETE trajectory of the fp32 run (`want`: its sampled bits are injected as
    uniforms 0 / 1 -- u < p gives the same bit at any precision) TWICE: from the parameters the fp32 ORACLE held before
    minibatch i (params_oracle[i]) and from the parameters the GPU held (params_gpu[i]; equal for i = 0, later they differ by
    the two implementations' own update noise -- RMSprop turns rounding-level gradients into +-lr steps).  Returns
    {"mb<i>.losses": (exact at the GPU's parameters, exact at the oracle's parameters)}; compare_packed gates
        |GPU - exact(GPU params)| <= |fp32 oracle - exact(oracle params)| + 1e-4
    i.e. each implementation against the exact value of ITS OWN forward pass: "at least as close to exact as the reference's
    fp32 arithmetic, plus north_star's 1e-4" (replaces round 4's 16-ulp allow-list)."""
    fl = flags_from_meta(meta)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    keys6 = ("nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen")
    try:
        out = {}
        for i in range(meta["n_minibatches"]):
            x, target, desc, (u_z, u_s, u_w) = case_inputs(meta, i, name)
            u = {"z": np.array(u_z, np.float64), "s": np.array(u_s, np.float64), "w": np.array(u_w, np.float64)}
            if fl.use_binary:
                for kind, key in (("z", "sen_feats"), ("s", "s_feats"), ("w", "rec_feats")):
                    bits = np.asarray(want["mb%d.%s" % (i, key)])                    # [n, B, .] of the executed steps
                    n = bits.shape[0]
                    u[kind][:n] = np.where(bits.reshape(u[kind][:n].shape) != 0, 0.0, 1.0)
            both = []
            for params in ((params_gpu[i], params_oracle[i]) if i > 0 else (params_oracle[i],)):
                tape = cpu_ref.UniformTape()
                models = cpu_ref.build_agents(fl, rng=tape)
                for a, m in models.items():
                    m.double()
                    m.load_state_dict({k: torch.as_tensor(v).double() for k, v in params[a].items()})
                optimizers = cpu_ref.build_optimizers(models, fl)
                tape.u = {k: v.copy() for k, v in u.items()}
                tape.t = {"z": 0, "s": 0, "w": 0}
                res = cpu_ref.train_minibatch(models, optimizers, torch.from_numpy(x).double(), torch.from_numpy(target),
                                              torch.from_numpy(desc).double(), fl, update=False)
                assert int(res["n_steps"]) == int(want["mb%d.n_steps" % i]), "the float64 re-run left the fp32 run's trajectory"
                both.append(np.array([float(res[k].detach()) for k in keys6], np.float64))
            out["mb%d.losses" % i] = (both[0], both[-1])
        return out
    finally:
        torch.set_default_dtype(old)


def forced_masks(flips, capture, already=None):
    """Per minibatch: the near-threshold units of the oracle run (flips[i]["pos"]) that the OTHER implementation put on the other
    side.  capture[i] = that implementation's own pre-activations for minibatch i: {"Astar": [B, R], "Cd": [D, R]} (the y head's
    pre-activation at the output step is Astar[b, r] + Cd[d, r]) and {"hid_r", "hid_s": [T, B, K]} (post-ReLU
    hidden units of the baselines).  already: positions forced in an earlier pass (kept, with the side they were given)."""
    out = []
    for i, f in enumerate(flips):
        cap = capture[i]
        forced = {p[:-1]: p[-1] for p in (already[i] if already else ())}
        for p in f["pos"]:
            if p[:-1] in forced:
                continue
            if p[0] == "y":
                _, t, b, d, r, side = p
                if "y_pre" in cap:                               # explicit per-(b, d, r) values (tests of the gate itself)
                    other = bool(cap["y_pre"][(b, d, r)] > 0)
                else:
                    other = bool(float(cap["Astar"][b, r]) + float(cap["Cd"][d, r]) > 0)
            else:
                which, t, b, k, side = p
                other = bool(float(cap["hid_r" if which == "bas_rec" else "hid_s"][t, b, k]) > 0)
            if other != side:
                forced[p[:-1]] = other
        out.append({k + (v,) for k, v in forced.items()})
    return out


def sampling_margin(packed, meta, name=None):
    """Smallest |u - p| over every Bernoulli draw of an oracle run (packed dict): a bit whose uniform lies within rounding
    distance of its probability may come out differently in another correct fp32 implementation, and then that whole
    conversation differs.  Tests with hundreds of thousands of draws pick seeds whose margin is comfortably above 1e-5."""
    m = float("inf")
    for i in range(meta["n_minibatches"]):
        _, _, _, (u_z, u_s, u_w) = case_inputs(meta, i, name)
        n = int(packed["mb%d.n_steps" % i])
        for key, u in (("sen_probs", u_z), ("s_probs", u_s), ("rec_probs", u_w)):
            p = np.asarray(packed["mb%d.%s" % (i, key)])
            if p.size:
                m = min(m, float(np.abs(np.asarray(u)[:n].reshape(p.shape) - p).min()))
    return m


def separate_draws(name, meta, margin=1e-4):
    """Registers uniforms for the case `name` in which no Bernoulli draw lies within `margin` of its probability: the oracle
    runs once on the seeded uniforms, every u with |u - p| < margin is moved to p -/+ margin on the side it was on (the
    sampled bit, hence the whole trajectory and every later p, is unchanged), and the adjusted arrays become the case's
    inputs (U_OVERRIDES).  With ~10^5 draws per minibatch some draw always sits within 1e-6 of p, where two correct fp32
    implementations toss a coin; this removes the coin, not the comparison."""
    U_OVERRIDES.pop(name, None)
    packed = oracle_train_case(name, meta)
    adjusted = []
    for i in range(meta["n_minibatches"]):
        _, _, _, us = case_inputs(meta, i, None)
        us = [np.array(u, copy=True) for u in us]
        n = int(packed["mb%d.n_steps" % i])
        for key, u in zip(("sen_probs", "s_probs", "rec_probs"), us):
            p = np.asarray(packed["mb%d.%s" % (i, key)], dtype=np.float64)
            if not p.size:
                continue
            v = u[:n].reshape(p.shape)                                    # a view: edits land in u
            close = np.abs(v - p) < margin
            v[close] = np.where(v[close] < p[close], p[close] - margin, p[close] + margin).astype(v.dtype)
            np.clip(v, 0.0, 0.99999994, out=v)
        adjusted.append(tuple(us))
    U_OVERRIDES[name] = lambda i, u_z, u_s, u_w: adjusted[i]
    return adjusted


SHIFT_INVARIANT = ("y", "outp")

# north_star: "logits/loss within 1e-4 of the reference CPU path".  Forward quantities -- logits, log-probabilities,
# probabilities, rewards, baseline scores and the six loss scalars -- are compared with an ABSOLUTE tolerance of 1e-4, whatever
# their magnitude.  The REINFORCE losses of config 4 are means of 256-bit log-likelihood sums times a reward weight,
# |loss| ~ 600, where ONE fp32 ulp is 6.1e-5 and the fp32 oracle itself moves by 4e-4 between hosts: a caller that passes
# `f64` (oracle_losses_f64: the oracle's losses re-computed in float64 on the same discrete trajectory, from each side's own
# parameters) gets those entries gated as |got - exact(got's parameters)| <= |fp32 oracle - exact(its parameters)| + 1e-4
# -- against the exact value, with the reference's own fp32 error as the allowance
# (round 6; rounds 4-5 allowed 16 ulp of |want| there, which only said "as noisy as fp32").  Gradients / updated parameters /
# gradient norms (sums over up to 10^4 products) keep atol + rtol.
FORWARD_ATOL = 1e-4
GRAD_KEYS = (".g.", ".p.", "gradnorm")

# max abs error per (case label, quantity) seen by compare_packed in this process; tests/conftest.py writes it to
# tests/out/parity_maxerr.json at the end of a GPU session (a copy is committed under profiles/)
MAXERR = {}


def is_grad_key(k):
    return any(t in k for t in GRAD_KEYS)


GATE = {}           # per (case label, quantity): which forward gate applied -- "abs 1e-4" or "f64: |got - f64| <= |fp32 oracle - f64| + 1e-4 (...)"


def compare_packed(got, want, atol=1e-5, rtol=1e-4, skip=(), only_prefix=None, shift_invariant=False, label=None, details=None, f64=None):
    """Compare two packed dicts.  Bit/mask/count entries must match exactly; forward float entries within
    min(atol, 1e-4) ABSOLUTE (FORWARD_ATOL) of `want` -- entries that `f64` holds (oracle_losses_f64) within
    |want - f64| + min(atol, 1e-4) of the float64 value instead; gradient / parameter entries within atol + rtol*|want|.

    shift_invariant: compare the class logits (``y``, ``outp``) of every minibatch AFTER the first one with each row's mean removed
    (``mb0.*``: raw -- no update has happened yet).
    dL/d(y2.bias) is identically zero (softmax is shift invariant), so what reaches the optimizer
    is rounding noise which RMSprop/Adam normalise into +-O(lr) steps: y2.bias (a common shift of
    all logits of a sample, invisible to every loss and to top-k) performs an implementation-
    dependent random walk in the reference too and cannot be pinned."""
    # details: a list that receives (key, offending flat indices, message) for every failing float entry
    exact = ("s_masks", "s_feats", "sen_feats", "rec_feats", "n_steps", "hits")
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
        # (minibatch 0 has seen no update: y2.bias is the initial one on both sides and the RAW logits must agree)
        if shift_invariant and tail in SHIFT_INVARIANT and a.size and not k.startswith("mb0."):
            a = a - a.mean(-1, keepdims=True)
            b = b - b.mean(-1, keepdims=True)
        is_bits = b.size == 0 or bool(np.all((b == 0) | (b == 1))) or tail in ("n_steps", "hits")
        if tail in exact and is_bits:      # continuous-mode messages are real-valued -> tolerance
            if not np.array_equal(a, b):
                problems.append("%s differs (exact) in %d places" % (k, int((a != b).sum())))
        else:
            err = np.abs(a.astype(np.float64) - b.astype(np.float64))
            gate64 = f64 is not None and k in f64 and not is_grad_key(k)
            if is_grad_key(k):
                tol = atol + rtol * np.abs(b.astype(np.float64))
            elif gate64:
                ex_got, ex_want = [np.asarray(v, np.float64).reshape(b.shape) for v in
                                   (f64[k] if isinstance(f64[k], tuple) else (f64[k], f64[k]))]
                err = np.abs(a.astype(np.float64) - ex_got)          # each side against the exact value of ITS OWN forward pass
                tol = np.abs(b.astype(np.float64) - ex_want) + min(atol, FORWARD_ATOL)
            else:
                tol = np.full(b.shape, min(atol, FORWARD_ATOL), dtype=np.float64)
            if a.size:
                if label is not None:
                    key = "%s:%s" % (label, k)
                    MAXERR[key] = max(MAXERR.get(key, 0.0), float(err.max()))
                    if not is_grad_key(k):
                        GATE[key] = ("f64: |got - f64| <= |fp32 oracle - f64| + 1e-4 (max |want| = %.3g, fp32 oracle off by %.1e)" % (
                            float(np.abs(b).max()), float((tol - min(atol, FORWARD_ATOL)).max()))) if gate64 else "abs 1e-4"
                if not np.all(err <= tol):
                    msg = "%s max err %.3e (tol %.1e)" % (k, float(err.max()), float(tol.flat[err.argmax()]))
                    problems.append(msg)
                    if details is not None:
                        details.append((k, np.nonzero((err > tol).reshape(-1))[0], msg))
    return problems


def param_shapes(eng):
    return {a: {k: tuple(v.shape) for k, v in d.items()} for a, d in eng.params.items()}


def assert_parity(got, want, flips, eng, label, skip=(), atol=1e-4, rtol=1e-3, max_passes=4, f64=None):
    """THE parity gate of the GPU tests.  Forward quantities (logits, probabilities, rewards, baseline scores, the six losses):
    |got - want| <= 1e-4 absolute (entries of `f64`, the oracle's float64 re-run: |got - f64| <= |want - f64| + 1e-4; GATE
    records which applied); bits / masks / counts exact.  Gradients, updated parameters, gradient norms: atol + rtol |want|.

    d relu/dx is discontinuous: a y-head or baseline hidden unit whose pre-activation is within RELU_EPS of zero may land on
    the other side in a correct fp32 implementation with another summation order, and then the gradients it feeds differ by
    O(dy * w2).  Such a mismatch is NOT excused: the oracle is re-run with exactly those units -- near-threshold in the
    oracle's own run AND on the other side on the GPU (eng.relu_capture: the GPU's own pre-activations, forced_masks) --
    forced to the GPU's side, and the re-run must agree with the GPU on EVERY compared entry within the normal tolerances
    (a forced unit moves the forward pass by < RELU_EPS * |w2|).  New near-threshold units of the re-run (its later
    minibatches start from slightly different parameters) are forced the same way, at most `max_passes` times.  A mismatch
    with no such unit, or one that survives the re-run, fails."""
    details = []
    problems = compare_packed(got, want, atol=atol, rtol=rtol, skip=skip, shift_invariant=True, label=label, details=details, f64=f64)
    # a forward mismatch fails at once -- unless an EARLIER minibatch has a gradient mismatch (its update then moved the
    # parameters this minibatch starts from): those are left to the forced re-run, which must clear them too
    mb_of = lambda p: int(p.split(".")[0][2:]) if p.startswith("mb") and p.split(".")[0][2:].isdigit() else -1
    grad_mbs = [mb_of(p) for p in problems if is_grad_key(p.split(" ")[0])]
    first_grad = min(grad_mbs) if grad_mbs else 1 << 30
    hard = [p for p in problems if not is_grad_key(p.split(" ")[0]) and mb_of(p) <= first_grad]
    assert not hard, "forward mismatch:\n" + "\n".join(hard[:20])
    if not problems:
        return problems
    capture = getattr(eng, "relu_capture", None)
    where = [w for f in flips for w in f["where"]][:12]
    assert capture is not None and flips and "case" in flips[0], \
        "gradient mismatch and no mask capture to re-run the oracle with:\n" + "\n".join(problems[:20])
    name, meta = flips[0]["case"]
    force, cur_flips, left = None, flips, problems
    for _ in range(max_passes):
        new_force = forced_masks(cur_flips, capture, already=force)
        if force is not None and all(a == b for a, b in zip(new_force, force)):
            break                                         # nothing new to force: the mismatch is real
        if force is None and not any(new_force):
            break                                         # no near-threshold unit sits on the other side on the GPU
        force = new_force
        cur_flips = []
        want2 = oracle_train_case(name, meta, flips=cur_flips, force=force)
        want2 = {k: want2[k] for k in (want.keys() if hasattr(want, "keys") else want.files) if k in want2}
        left = compare_packed(got, want2, atol=atol, rtol=rtol, skip=skip, shift_invariant=True,
                              label=(label + "/forced") if label else None, f64=f64)
        if not left:
            return problems
    n_forced = sum(len(f) for f in force) if force else 0
    assert False, ("gradient entries that differ from the oracle, also with the oracle's %d near-threshold ReLU units forced "
                   "to the GPU's side:\n" % n_forced) + "\n".join(left[:20]) + "\nnear-threshold units: " + "; ".join(where)


def relu_margin(eng):
    """Smallest |pre-activation| of a ReLU unit that carries gradient in the engine's last minibatch: the y head's
    A*[b, r] + Cd[d, r] over classes with |dy| > 0 and the baselines' hidden units that are just above zero.  A
    gradient entry may disagree with the CPU oracle only through a mask flip of such a unit (d relu/dx is
    discontinuous), so a test that excuses a gradient mismatch must find this margin below 1e-5."""
    tp = eng.tape
    A, Cd, dy = tp["Astar"].double().cpu(), tp["Cd"].double().cpu(), tp["dy"].double().cpu()
    pre = (A[:, None, :] + Cd[None, :, :]).abs()
    pre[(dy.abs() == 0)[:, :, None].expand_as(pre)] = float("inf")
    m = float(pre.min())
    for name in ("hid_s", "hid_r"):
        h = tp[name].double().cpu()
        pos = h[h > 0]
        if pos.numel():
            m = min(m, float(pos.min()))
    return m


# ----------------------------------------------------------------------------------------------
# HIP path driven through the C-ABI (GPU tests only)
# ----------------------------------------------------------------------------------------------
class _SD(object):
    def __init__(self, d):
        self._d = d

    def state_dict(self):
        return self._d


def engine_kwargs(meta, batch=None, **over):
    fl = flags_from_meta(meta)
    kw = dict(batch=batch or meta["batch"], n_classes=meta["n_classes"], feat_dim=fl.img_feat_dim, h_dim=fl.img_h_dim,
              w_dim=fl.rec_w_dim, rec_hidden=fl.rec_hidden, wv_dim=fl.wv_dim, bas_hidden=fl.baseline_hid_dim,
              max_exchange=fl.max_exchange, use_binary=fl.use_binary, fixed_exchange=fl.fixed_exchange,
              s_prob_prod=fl.s_prob_prod, entropy_s=fl.entropy_s, entropy_sen=fl.entropy_sen,
              entropy_rec=fl.entropy_rec, first_rec=fl.first_rec, optim_type=fl.optim_type,
              learning_rate=fl.learning_rate, top_k=fl.top_k_train)
    kw.update(over)
    return kw


def make_engine(meta, **over):
    from multimodalgame_amd.engine import Engine
    eng = Engine(**engine_kwargs(meta, **over))
    shapes = {a: {k: tuple(v.shape) for k, v in d.items()} for a, d in eng.params.items()}
    eng.load_state_dicts(cpu_ref.fill_state_dicts(shapes, seed=meta["seed_weights"]))
    return eng


def engine_result(eng, fl):
    """Slice the tape the way exchange() returns it (lists over the executed steps)."""
    torch.cuda.synchronize()
    eng.check_sync()
    tp = {k: v.cpu() for k, v in eng.tape.items() if k in (
        "mask", "s", "ps", "z", "pz", "w", "pw", "y", "bs", "br", "outp", "dist", "logs", "losses")}
    losses = tp["losses"].tolist()
    n = int(losses[6])
    binary = fl.use_binary
    masks = [tp["mask"][t].clone() for t in range(n + 1)]
    masks[-1].zero_()                                                    # model.py:870
    res = dict(n_steps=n, s_masks=masks,
               s_feats=[tp["s"][t] for t in range(n)], s_probs=[tp["ps"][t] for t in range(n)],
               sen_feats=[tp["z"][t] for t in range(n)], sen_probs=[tp["pz"][t] if binary else None for t in range(n)],
               rec_feats=[tp["w"][t] for t in range(n)], rec_probs=[tp["pw"][t] if binary else None for t in range(n)],
               y=[tp["y"][t] for t in range(n)], bs=[tp["bs"][t] for t in range(n)], br=[tp["br"][t] for t in range(n)],
               outp=tp["outp"], dist=tp["dist"], logs=tp["logs"].view(-1, 1), hits=int(losses[7]))
    for i, k in enumerate(("nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen")):
        res[k] = torch.tensor(losses[i])
    return res


def hip_train_case(name, meta, early_exit=False, fused=False):
    """Run a golden train case on the GPU through the C-ABI; returns (packed dict, engine)."""
    fl = flags_from_meta(meta)
    eng = make_engine(meta)
    dev = eng.device
    out = {}
    agents = ("receiver", "sender", "baseline_rec", "baseline_sen") if fl.use_binary else ("receiver",)
    eng.relu_capture = []
    eng.param_snapshots = []                                        # the parameters every minibatch starts from (oracle_losses_f64)
    for i in range(meta["n_minibatches"]):
        eng.param_snapshots.append({a: {k: v.detach().cpu().clone() for k, v in eng.params[a].items()} for a in _lib_agents()})
        x, target, desc, (u_z, u_s, u_w) = case_inputs(meta, i, name)
        xd, td, dd = torch.from_numpy(x).to(dev), torch.from_numpy(target).to(dev), torch.from_numpy(desc).to(dev)
        uz, us, uw = [torch.from_numpy(np.ascontiguousarray(u)).to(dev) for u in (u_z, u_s[..., 0], u_w)]
        if fused:
            eng.train_step(xd, td, dd, uz, us, uw)
        else:
            eng.forward(xd, td, dd, uz, us, uw, train=True, run_all=not early_exit)
            eng.loss_stats()
            eng.backward(xd, td, dd)
        res = engine_result(eng, fl)
        # the GPU's own ReLU pre-activations of this minibatch (assert_parity / forced_masks): y head at the output step =
        # Astar + Cd, baselines = their stored hidden units
        eng.relu_capture.append({k: eng.tape[k].detach().cpu().clone() for k in
                                 (("Astar", "Cd", "hid_r", "hid_s") if fl.use_binary else ("Astar", "Cd"))})
        grads, norms = {}, {}
        for a in agents:
            grads[a] = {k: v.detach().cpu().clone() for k, v in eng.grads[a].items()}
            lo, hi = eng.agent_range[a]
            norms[a] = float(eng.flat_grads[lo:hi].double().norm())
        res["grads"], res["grad_norms"] = grads, norms
        if not fused:
            eng.clip_step()
        torch.cuda.synchronize()
        models = {a: _SD({k: v.detach().cpu() for k, v in eng.params[a].items()}) for a in _lib_agents()}
        out.update(cpu_ref.pack_train(res, models, prefix="mb%d." % i))
    return out, eng


def _lib_agents():
    return ("receiver", "sender", "baseline_rec", "baseline_sen")
