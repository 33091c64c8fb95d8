"""Parity of the HIP path with the CPU oracle at the shapes of BASELINE.json configs[3] and [4]
(generic kernels: dimensions outside the register-resident specialisation), plus size-independent
properties at config-5's full size where the oracle (1.87 GB of build_inp activations per step) is too slow."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from tests import common

pytestmark = pytest.mark.gpu

C4 = dict(use_binary=True, fixed_exchange=False, max_exchange=10, batch_size=64, learning_rate=1e-4, entropy_rec=0.01,
          entropy_sen=0.01, entropy_s=0.08, img_feat_dim=512, img_h_dim=1024, rec_w_dim=256, sender_out_dim=256,
          rec_hidden=64, wv_dim=100, baseline_hid_dim=500, top_k_train=6)
C5 = dict(use_binary=False, fixed_exchange=True, max_exchange=10, batch_size=128, learning_rate=1e-4,
          img_feat_dim=512, img_h_dim=256, rec_w_dim=32, sender_out_dim=32, rec_hidden=64, wv_dim=100,
          baseline_hid_dim=500, top_k_train=6)


def _meta(flags_kw, n_classes, batch, n_mb, seeds=(5, 6, 7)):
    fl = cpu_ref.Flags(**flags_kw)
    meta = dict(fl.__dict__)
    meta.update(n_classes=n_classes, batch=batch, n_minibatches=n_mb, seed_weights=seeds[0], seed_data=seeds[1],
                seed_uniforms=seeds[2])
    return meta


_ORACLE_CACHE = {}


def _oracle(meta, key=None, f64=False, eng=None):
    """(want, flips, f64 gate): the oracle's fp32 run of the case (cached per session under `key`: it takes seconds at config 4)
    and -- f64=True, the cases with config 4's 256-bit agents -- the float64 values of the six losses on the same discrete
    trajectory from the oracle's AND from the GPU's own parameters (common.oracle_losses_f64; eng: the engine
    common.hip_train_case returned), against which the losses are gated there."""
    if key is not None and key in _ORACLE_CACHE:
        want, flips, params = _ORACLE_CACHE[key]
    else:
        flips, params = [], []
        want = common.oracle_train_case(None, meta, flips=flips, params_before=params)
        if key is not None:
            _ORACLE_CACHE[key] = (want, flips, params)
    return want, flips, (common.oracle_losses_f64(None, meta, want, params, eng.param_snapshots) if f64 else None)


def _compare(meta, skip, label, key=None):
    """Forward quantities and losses: the 1e-4 gate of common.assert_parity (config 4's losses: against the float64 oracle).  A
    gradient entry beyond its tolerance passes only if a ReLU unit on the threshold (|pre| < RELU_EPS in the oracle's own run)
    feeds it."""
    got, eng = common.hip_train_case(None, meta)
    want, flips, f64 = _oracle(meta, key, f64=label.startswith("config4"), eng=eng)
    common.assert_parity(got, want, flips, eng, label, skip=skip, f64=f64)


def _compare_cached(meta, skip, label, key):
    _compare(meta, skip, label, key)


@pytest.mark.parametrize("switch", [None, "MMG_NO_PERSIST_LL", "MMG_NO_FUSED_S", "MMG_NO_RMSG", "MMG_NO_RSAMPLE", "MMG_NO_PERSIST"])
def test_config4_shape_vs_oracle(switch, monkeypatch):
    """Adaptive, W=256, H=1024 (configs[3]): 1 952 852 parameters.  Default: one persistent launch of per-sample receiver
    roles + fused sender roles that hand over (value, epoch) pairs; MMG_NO_PERSIST_LL: the same roles with payload + counter
    hand-offs.  The other switches walk down the fallbacks other agent shapes take: hidden-slice / bit-slice
    sender roles (s1 / s2), receiver roles that publish g instead of the message, one 16-sample receiver role per tile,
    and three launches per step."""
    if switch:
        monkeypatch.setenv(switch, "1")
    _compare_cached(_meta(C4, 30, 64, 2), skip=("y2.bias",), label="config4" + ("-" + switch if switch else ""), key="c4")


@pytest.mark.parametrize("switch", [None, "MMG_NO_PERSIST_LL", "MMG_NO_FUSED_S", "MMG_NO_RMSG", "MMG_NO_RSAMPLE", "MMG_NO_PERSIST"])
def test_config4_fused_train_step_vs_oracle(switch, monkeypatch):
    """EXACTLY what bench.py's c4 line runs: the FUSED mmg_train_step with early stopping (samples leave their tile's ring of roles
    at different steps) on config 4's agents, two minibatches against the oracle.  test_config4_shape_vs_oracle runs every
    sample through every step (run_all), where no role ever has to tell a stopped sample from a slow one -- round 5 found the
    counter hand-offs of rounds 3-4 letting the sender roles run ahead of live samples once others had stopped."""
    if switch:
        monkeypatch.setenv(switch, "1")
    meta = _meta(dict(C4), 30, 64, 2)
    got, eng = common.hip_train_case(None, meta, fused=True)
    want, flips, f64 = _oracle(meta, "c4-fused", f64=True, eng=eng)
    common.assert_parity(_pick(got), _pick(want), flips, eng, "config4-fused" + ("-" + switch if switch else ""), skip=("y2.bias",), f64=f64)


@pytest.mark.parametrize("batch", [24, 88])
def test_config4_fused_train_step_ragged_and_chunked(batch):
    """The same with a ragged last tile (24 samples: rows 8-15 of tile 1 have no receiver role) and with more roles than fit one
    launch (88 samples: two launches over tile ranges, the pair slots of the other range untouched), early stopping on."""
    meta = _meta(dict(C4, batch_size=batch), 30, batch, 2)
    got, eng = common.hip_train_case(None, meta, fused=True)
    want, flips, f64 = _oracle(meta, f64=True, eng=eng)
    common.assert_parity(_pick(got), _pick(want), flips, eng, "config4-fused-b%d" % batch, skip=("y2.bias",), f64=f64)


@pytest.mark.parametrize("switch", [None, "MMG_NO_RC_PERSIST", "MMG_NO_RC_BWD"])
def test_config4_with_rec_hidden_256_vs_oracle(switch, monkeypatch):
    """'rec_w_dim 256 / img_h_dim 1024 ... use R = 64 and additionally report R = 256'.  At R = 256 the
    one-workgroup-per-tile forward does not fit its LDS plan: the receiver of a tile is split over workgroups by 16-unit slices on
    the matrix cores (kernels_rc.h) -- one launch of co-resident roles (k_rc_persist), or with the switch k_rc_gru / k_rc_heads /
    k_rc_query between the per-step sender launches; the backward on the tile kernels, its reverse-time loop as roles over
    16-unit slices (k_rc_bwd, which also runs the output-step prelude per slice; MMG_NO_RC_BWD: all of it inside k_bwd_tile);
    B = 64 as the bench times it."""
    if switch:
        monkeypatch.setenv(switch, "1")
    meta = _meta(dict(C4, rec_hidden=256, batch_size=64), 30, 64, 2)
    got, eng = common.hip_train_case(None, meta)
    want, flips, f64 = _oracle(meta, "c4-R256", f64=True, eng=eng)
    common.assert_parity(got, want, flips, eng, "config4-R256", skip=("y2.bias",), f64=f64)
    names = _kernel_names(eng, meta)
    assert "k_conv_rc" in names and "k_bwd_tile" in names, names


@pytest.mark.parametrize("flavour", ["adaptive", "fixed", "continuous"])
def test_wide_receiver_fused_train_step_vs_oracle(flavour):
    """EXACTLY what bench.py's c4r256 line runs: the FUSED mmg_train_step (run_all_steps = 2: early stopping, live rows only, in
    Fixed mode y of the output step only, in continuous mode the lean tape) on the wide-receiver roles -- k_rc_persist, k_bwd_pre_send,
    k_rc_bwd, k_wgrad, k_opt -- two minibatches at B = 64 against the oracle."""
    kw = dict(C4, rec_hidden=256, batch_size=64)
    skip = ("y2.bias",)
    if flavour == "fixed":
        kw.update(fixed_exchange=True, max_exchange=4)
    elif flavour == "continuous":
        kw.update(use_binary=False, fixed_exchange=True, max_exchange=4)
        skip = ("y2.bias", ".bs", ".br")
    meta = _meta(kw, 30, 64, 2)
    got, eng = common.hip_train_case(None, meta, fused=True)
    want, flips, f64 = _oracle(meta, f64=True, eng=eng)
    common.assert_parity(_pick(got), _pick(want), flips, eng, "config4-wide-fused-" + flavour, skip=skip, f64=f64)
    names = _kernel_names(eng, meta)
    assert "k_conv_rc" in names and "k_bwd_tile" in names and "k_conversation" not in names, names


@pytest.mark.parametrize("flavour", ["fixed", "continuous", "ragged", "r192", "b88"])
def test_wide_receiver_other_modes_vs_oracle(flavour):
    """The wide-receiver kernels (kernels_rc.h) outside config 4's own mode: Fixed exchange (every row live, output at T - 1),
    continuous messages (no sampling, receiver-only backward without k_bwd_pre), a ragged last tile (B = 24), rec_hidden 192
    (12 slices, three of the four waves hold a fourth k-group less) and 88 samples (more roles than fit the device at once: two
    consecutive launches over tile ranges)."""
    kw = dict(C4, rec_hidden=256, batch_size=32)
    B = 32
    skip = ("y2.bias",)
    if flavour == "fixed":
        kw.update(fixed_exchange=True, max_exchange=4)
    elif flavour == "continuous":
        kw.update(use_binary=False, fixed_exchange=True, max_exchange=4)
        skip = ("y2.bias", ".bs", ".br")
    elif flavour == "ragged":
        kw.update(batch_size=24); B = 24
    elif flavour == "b88":                       # 6 tiles (the last one ragged): two consecutive role launches over tile ranges (3 + 3)
        kw.update(batch_size=88); B = 88
    else:
        kw.update(rec_hidden=192)
    meta = _meta(kw, 30, B, 2)
    got, eng = common.hip_train_case(None, meta)
    want, flips, f64 = _oracle(meta, f64=True, eng=eng)
    common.assert_parity(got, want, flips, eng, "config4-wide-" + flavour, skip=skip, f64=f64)      # (config 4's 256-bit agents: the six losses against the float64 oracle)
    assert "k_conv_rc" in _kernel_names(eng, meta)


def test_wide_receiver_eval_pass_agrees_with_generic_kernels(monkeypatch):
    """Evaluation pass (train = False: rounded bits, running product of the stop probabilities) of the wide-receiver roles against
    the generic per-sample kernels on the same weights and inputs.  Rounding sigmoid outputs near 0.5 may differ between two
    correct fp32 summation orders and then changes everything downstream, so the comparison follows each sample up to its first
    such near-tie: all probabilities within 1e-4, bits equal away from ties."""
    meta = _meta(dict(C4, rec_hidden=256, batch_size=32), 30, 32, 1)
    x, target, desc, _ = common.case_inputs(meta, 0)

    def run():
        eng = common.make_engine(meta)
        dev = eng.device
        eng.forward(torch.from_numpy(x).to(dev), torch.from_numpy(target).to(dev), torch.from_numpy(desc).to(dev), train=False, run_all=True)
        torch.cuda.synchronize()
        eng.check_sync()
        return {k: v.cpu().numpy().copy() for k, v in eng.tape.items() if k in ("s", "ps", "z", "pz", "w", "pw", "y", "tstar", "dist")}
    got = run()
    monkeypatch.setenv("MMG_NO_RC", "1")
    want = run()
    T, B = got["ps"].shape[0], got["ps"].shape[1]
    ok = np.ones(B, bool)                                # samples whose conversations are still tie-free
    checked = 0
    for t in range(T):
        for pk, bk in (("pz", "z"), ("ps", "s")):
            np.testing.assert_allclose(got[pk][t][ok], want[pk][t][ok], atol=1e-4, err_msg="%s[%d]" % (pk, t))
            tie = (np.abs(want[pk][t].reshape(B, -1) - 0.5) < 2e-5).any(1)
            ok &= ~tie
            np.testing.assert_array_equal(got[bk][t][ok], want[bk][t][ok], err_msg="%s[%d]" % (bk, t))
        np.testing.assert_allclose(got["y"][t][ok] - got["y"][t][ok].mean(-1, keepdims=True),
                                   want["y"][t][ok] - want["y"][t][ok].mean(-1, keepdims=True), atol=1e-4, err_msg="y[%d]" % t)
        np.testing.assert_allclose(got["pw"][t][ok], want["pw"][t][ok], atol=1e-4, err_msg="pw[%d]" % t)
        ok &= ~(np.abs(want["pw"][t].reshape(B, -1) - 0.5) < 2e-5).any(1)
        np.testing.assert_array_equal(got["w"][t][ok], want["w"][t][ok], err_msg="w[%d]" % t)
        checked += int(ok.sum())
    assert checked >= B // 2, "too few tie-free (step, sample) rows were compared: %d" % checked
    np.testing.assert_array_equal(got["tstar"][ok], want["tstar"][ok])


def test_config4_with_rec_hidden_256_generic_fallback(monkeypatch):
    """MMG_NO_RC=1: the same shape on the generic per-sample kernels (what a device without the tile path's alignment runs)."""
    monkeypatch.setenv("MMG_NO_RC", "1")
    _compare(_meta(dict(C4, rec_hidden=256, batch_size=16), 30, 16, 2), skip=("y2.bias",), label="config4-R256-generic")


def test_config4_shape_consecutive_role_launches():
    """Config 4's agents at 88 samples: 6 tiles (the last one ragged) do not fit one launch of co-resident roles (240
    workgroups), so the conversation runs as two launches over sample ranges (48 + 40 samples)."""
    _compare(_meta(dict(C4, batch_size=88), 30, 88, 2), skip=("y2.bias",), label="config4-b88")


@pytest.mark.parametrize("kernels", ["default", "tile", "tile-nosplit"])
def test_config5_flavour_vs_oracle(kernels, monkeypatch):
    """1000 classes, continuous messages, Fixed (configs[4]), PHASED calls with every step's arrays kept (run-all).  "default":
    k_conversation_mc + k_bwd_mc1/2 (what a GPU's shard runs; the fused lean-tape step bench.py times is
    test_config5_shard_fused_vs_oracle); "tile" forces the sample-tile kernels with class helpers (k_conv_split);
    "tile-nosplit" the many-class y head of one workgroup (4 samples x 4 classes register blocks), the path of > 2 048
    samples per GPU."""
    if kernels.startswith("tile"):
        monkeypatch.setenv("MMG_TILE", "1")
    if kernels == "tile-nosplit":                   # what 2 048 samples per GPU run: every tile takes all 1000 classes itself
        monkeypatch.setenv("MMG_NO_SPLIT", "1")
    _compare(_meta(C5, 1000, 128, 2), skip=("y2.bias", ".bs", ".br"), label="config5-" + kernels)


KEEP_TRAIN = ("losses", "n_steps", "hits", "logs", "outp", "dist", ".g.", ".p.", "gradnorm")


def _pick(d, keep=KEEP_TRAIN):
    return {k: d[k] for k in (d.keys() if hasattr(d, "keys") else d.files) if any(t in k for t in keep)}


def _kernel_names(eng, meta, fused=True):
    x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, 0)
    dev = eng.device
    eng.set_profiling(True)
    eng.train_step(torch.from_numpy(x).to(dev), torch.from_numpy(target).to(dev), torch.from_numpy(desc).to(dev), seed=1)
    torch.cuda.synchronize()
    names = [n for n, _ in eng.kernel_times()]
    eng.set_profiling(False)
    return names


def test_config5_shard_fused_vs_oracle():
    """EXACTLY what bench.py's c5 line and each GPU of the 8-GPU job run (configs[4]: D = 1000, continuous, Fixed, 256 samples
    per GPU): the FUSED mmg_train_step -- lean tape (run_all_steps = 2: y of the output step only, no a / c / zr / dbar / g /
    w), k_conversation_mc, k_bwd_mc1, k_bwd_mc2 with the statistics workgroup riding along, k_wgrad (row slices added up by the last one to arrive), k_opt --
    two minibatches against the oracle (model.py:1297-1305, 1313: only the receiver is trained, loss = NLL)."""
    meta = _meta(dict(C5, batch_size=256), 1000, 256, 2)
    got, eng = common.hip_train_case(None, meta, fused=True)
    flips = []
    want = common.oracle_train_case(None, meta, flips=flips)
    common.assert_parity(_pick(got), _pick(want), flips, eng, "config5-shard-fused", skip=("y2.bias", ".bs", ".br"))
    names = _kernel_names(eng, meta)
    assert "k_conversation_mc" in names and "k_bwd_mc" in names and "k_stats" not in names, names


@pytest.mark.parametrize("sampling", ["injected", "philox"])
def test_config5_sharded_equals_unsharded(sampling):
    """configs[4] is the 8-GPU workload: the many-class continuous path keys its Philox streams on the GLOBAL sample index
    (dm.boff) and normalises the NLL by the GLOBAL batch (dm.Bg).  Two shards of 128 (batch_offset 0 / 128, global_batch 256)
    through the phased calls, statistics and gradients summed by hand as the data-parallel step does, must give the single
    256-sample engine's update, losses and rewards."""
    meta = _meta(dict(C5, batch_size=256), 1000, 256, 1)
    x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, 0)
    full = common.make_engine(meta)
    dev = full.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    inj = sampling == "injected"
    u = lambda a, sl: t(a[:, sl]) if inj else None
    full.forward(t(x), t(target), t(desc), u(u_z, slice(None)), u(u_s[..., 0], slice(None)), u(u_w, slice(None)), seed=7, train=True, run_all=False, minimal=True)
    full.loss_stats()
    full.backward(t(x), t(target), t(desc))
    torch.cuda.synchronize()
    g_full = full.flat_grads.clone()
    full.clip_step()
    shards = [common.make_engine(meta, batch=128, global_batch=256, batch_offset=128 * r) for r in range(2)]
    args = []
    for r, e in enumerate(shards):
        sl = slice(128 * r, 128 * r + 128)
        a = (t(x[sl]), t(target[sl]), t(desc), u(u_z, sl), u(u_s[..., 0], sl), u(u_w, sl))
        args.append(a)
        e.forward(*a, seed=7, train=True, run_all=False, minimal=True)
        e.loss_stats()
    stats = sum(e.stats.clone() for e in shards)                      # the all-reduce of dist.py, by hand
    for e, a in zip(shards, args):
        e.stats.copy_(stats)
        e.backward(a[0], a[1], a[2])
    torch.cuda.synchronize()
    grads = sum(e.flat_grads.clone() for e in shards)
    lo, hi = full.agent_range["receiver"]
    np.testing.assert_allclose(grads[lo:hi].cpu().numpy(), g_full[lo:hi].cpu().numpy(), rtol=2e-4, atol=2e-6)
    shards[0].flat_grads.copy_(grads)
    shards[0].clip_step()
    torch.cuda.synchronize()
    for r, e in enumerate(shards):                                    # rewards / selected logits of the shard's rows
        sl = slice(128 * r, 128 * r + 128)
        np.testing.assert_allclose(e.tape["logs"].cpu().numpy(), full.tape["logs"][sl].cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(e.tape["outp"].cpu().numpy(), full.tape["outp"][sl].cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(e.tape["s"].cpu().numpy(), full.tape["s"][:, sl].cpu().numpy())       # sampled stop bits
    for k, v in full.params["receiver"].items():
        if k == "y2.bias":
            continue
        a, b = shards[0].params["receiver"][k].cpu().numpy(), v.cpu().numpy()
        bad = ~np.isclose(a, b, rtol=2e-4, atol=2e-6)
        # elements whose gradient is rounding noise take an RMSprop-normalised step whose sign depends on the summation order
        assert bad.mean() <= 1e-4 and np.abs(a - b).max() <= 2 * 10 * 1e-4, "receiver.%s: %d bad" % (k, bad.sum())
    np.testing.assert_allclose(shards[0].tape["losses"][:1].cpu().numpy(), full.tape["losses"][:1].cpu().numpy(), rtol=1e-5, atol=1e-6)
    names = _kernel_names(common.make_engine(meta, batch=128), dict(meta, batch=128))
    assert "k_conversation_mc" in names and "k_bwd_mc" in names, names


def test_config5_full_size_fused_vs_oracle():
    """configs[4] at its FULL size on one GPU -- B = 2048, D = 1000, continuous, Fixed: what bench.py's `c5s` line (and the N = 1 point
    of --scaling strong) times.  The path differs from the 256-sample shard's in a way that matters: 2 048 role workgroups run as 8
    rounds of 256 with in-launch hand-offs inside each 16-member tile.  Reference semantics: build_inp over B * D = 2.05 M rows
    (model.py:519-551) and the y head / softmax . desc (model.py:432-449); the oracle materialises them (1.34 GB per exchange step plus
    autograd copies), so the conversation is cut to max_exchange = 3 -- every per-step code path of the kernels (t = 0, a middle step,
    the output step) still runs, in the same 8 rounds.  Same gate as test_config5_shard_fused_vs_oracle."""
    meta = _meta(dict(C5, batch_size=2048, max_exchange=3), 1000, 2048, 1)
    got, eng = common.hip_train_case(None, meta, fused=True)
    flips = []
    want = common.oracle_train_case(None, meta, flips=flips)
    common.assert_parity(_pick(got), _pick(want), flips, eng, "config5-full-fused", skip=("y2.bias", ".bs", ".br"))
    names = _kernel_names(eng, meta)
    assert "k_conversation_mc" in names and "k_bwd_mc" in names and "k_stats" not in names, names


def test_config5_full_size_properties():
    """B=2048 per call, D=1000, continuous: (1) bitwise-reproducible, (2) the loss goes down over a few
    updates on a fixed batch, (3) NLL equals -mean of the stored per-sample rewards, (4) selected logits are
    the last step's (Fixed)."""
    meta = _meta(dict(C5, batch_size=2048, learning_rate=1e-3), 1000, 2048, 1)
    x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, 0)
    outs = []
    for rep in range(2):
        eng = common.make_engine(meta)
        dev = eng.device
        xd, td, dd = [torch.from_numpy(a).to(dev) for a in (x, target, desc)]
        us = torch.from_numpy(np.ascontiguousarray(u_s[..., 0])).to(dev)
        losses = []
        for it in range(6):
            eng.train_step(xd, td, dd, None, us, None)
            losses.append(eng.losses()["nll_loss"])
        torch.cuda.synchronize()
        outs.append((losses, eng.flat_params.cpu().numpy().copy(), eng.tape["logs"].cpu().numpy().copy(),
                     eng.tape["outp"].cpu().numpy().copy(), eng.tape["y"][-1].cpu().numpy().copy()))
    np.testing.assert_array_equal(outs[0][1], outs[1][1])            # deterministic: no float atomics anywhere
    losses, _, logs, outp, y_last = outs[0]
    assert losses[-1] < losses[0], losses
    np.testing.assert_allclose(losses[-1], -logs.mean(), rtol=1e-5)
    np.testing.assert_array_equal(outp, y_last)


def test_config3_global_batch_512_fixed_sharded_equals_unsharded():
    """configs[2]: Fixed, global batch 512.  Eight shards of 64 computed one after another on this GPU with the
    DP protocol's statistics / gradient sums done by hand must give the single-engine B=512 update."""
    fl_kw = dict(use_binary=True, fixed_exchange=True, max_exchange=10, batch_size=512, learning_rate=1e-4,
                 entropy_rec=0.01, entropy_sen=0.01, img_feat_dim=512, img_h_dim=256, rec_w_dim=32, sender_out_dim=32,
                 rec_hidden=64, wv_dim=100, baseline_hid_dim=500, top_k_train=6)
    meta = _meta(fl_kw, 30, 512, 1)
    x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, 0)
    full = common.make_engine(meta)
    dev = full.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    full.train_step(t(x), t(target), t(desc), t(u_z), t(u_s[..., 0]), t(u_w))
    torch.cuda.synchronize()
    shards = [common.make_engine(meta, batch=64, global_batch=512, batch_offset=64 * r) for r in range(8)]
    args = []
    for r, e in enumerate(shards):
        sl = slice(64 * r, 64 * r + 64)
        a = (t(x[sl]), t(target[sl]), t(desc), t(u_z[:, sl]), t(u_s[:, sl, 0]), t(u_w[:, sl]))
        args.append(a)
        e.forward(*a, train=True, run_all=False)
        e.loss_stats()
    stats = sum(e.stats.clone() for e in shards)                      # the all-reduce of dist.py, by hand
    for e, a in zip(shards, args):
        e.stats.copy_(stats)
        e.backward(a[0], a[1], a[2])
    grads = sum(e.flat_grads.clone() for e in shards)
    shards[0].flat_grads.copy_(grads)
    shards[0].clip_step()
    torch.cuda.synchronize()
    for agent, d in full.params.items():
        for k, v in d.items():
            if k == "y2.bias":
                continue
            a, b = shards[0].params[agent][k].cpu().numpy(), v.cpu().numpy()
            bad = ~np.isclose(a, b, rtol=2e-4, atol=2e-6)
            # elements whose gradient is rounding noise take an RMSprop-normalised step whose sign depends on the
            # summation order (|step| <= lr / sqrt(1 - alpha) = 10 lr): allow a vanishing fraction of those
            assert bad.mean() <= 1e-4 and np.abs(a - b).max() <= 2 * 10 * 1e-4, "%s.%s: %d bad" % (agent, k, bad.sum())
    np.testing.assert_allclose(shards[0].tape["losses"][:6].cpu().numpy(), full.tape["losses"][:6].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_wide_receiver_sharded_equals_unsharded():
    """The wide-receiver roles (kernels_rc.h) under the data-parallel protocol: Adaptive, in-kernel Philox keyed on the GLOBAL
    sample index (dm.boff), NLL normalised by the GLOBAL batch (dm.Bg).  Two shards of 32 (two tiles each) computed one after another with the statistics / gradient sums done by hand must reproduce the single 64-sample engine:
    sampled bits and rewards exactly, the update within the noise-step allowance of the config-3 test."""
    meta = _meta(dict(C4, rec_hidden=256, batch_size=64), 30, 64, 1)
    x, target, desc, _ = common.case_inputs(meta, 0)
    full = common.make_engine(meta)
    dev = full.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    full.forward(t(x), t(target), t(desc), seed=11, train=True, run_all=False)
    full.loss_stats()
    full.backward(t(x), t(target), t(desc))
    full.clip_step()
    torch.cuda.synchronize()
    shards = [common.make_engine(meta, batch=32, global_batch=64, batch_offset=32 * r) for r in range(2)]
    args = []
    for r, e in enumerate(shards):
        sl = slice(32 * r, 32 * r + 32)
        a = (t(x[sl]), t(target[sl]), t(desc))
        args.append(a)
        e.forward(*a, seed=11, train=True, run_all=False)
        e.loss_stats()
    stats = sum(e.stats.clone() for e in shards)                      # the all-reduce of dist.py, by hand
    for e, a in zip(shards, args):
        e.stats.copy_(stats)
        e.backward(*a)
    grads = sum(e.flat_grads.clone() for e in shards)
    shards[0].flat_grads.copy_(grads)
    shards[0].clip_step()
    torch.cuda.synchronize()
    for r, e in enumerate(shards):
        e.check_sync()
        sl = slice(32 * r, 32 * r + 32)
        ts_f, ts_s = full.tape["tstar"][sl].cpu().numpy(), e.tape["tstar"].cpu().numpy()
        np.testing.assert_array_equal(ts_s, ts_f)
        np.testing.assert_allclose(e.tape["logs"].cpu().numpy(), full.tape["logs"][sl].cpu().numpy(), rtol=1e-5, atol=1e-5)
        for k in ("s", "z", "w"):                                       # live rows: t <= t* (w: t < t*)
            a, b = e.tape[k].cpu().numpy(), full.tape[k][:, sl].cpu().numpy()
            for bi in range(32):
                n = int(ts_f[bi]) + (0 if k == "w" else 1)
                np.testing.assert_array_equal(a[:n, bi], b[:n, bi], err_msg="%s sample %d" % (k, bi))
    for agent, d in full.params.items():
        for k, v in d.items():
            if k == "y2.bias":
                continue
            a, b = shards[0].params[agent][k].cpu().numpy(), v.cpu().numpy()
            bad = ~np.isclose(a, b, rtol=2e-4, atol=2e-6)
            assert bad.mean() <= 1e-4 and np.abs(a - b).max() <= 2 * 10 * 1e-4, "%s.%s: %d bad" % (agent, k, bad.sum())
    np.testing.assert_allclose(shards[0].tape["losses"][:6].cpu().numpy(), full.tape["losses"][:6].cpu().numpy(), rtol=2e-5, atol=1e-5)


def test_config3_global_batch_512_single_engine_vs_oracle():
    """configs[2] as the N = 1 point of strong scaling runs it: all 512 samples on ONE engine (k_baselines2, row-split k_wgrad
    -- paths the 64-sample shards never take), two minibatches, against the CPU oracle itself."""
    fl_kw = dict(use_binary=True, fixed_exchange=True, max_exchange=10, batch_size=512, learning_rate=1e-4,
                 entropy_rec=0.01, entropy_sen=0.01, img_feat_dim=512, img_h_dim=256, rec_w_dim=32, sender_out_dim=32,
                 rec_hidden=64, wv_dim=100, baseline_hid_dim=500, top_k_train=6)
    # 164 k Bernoulli draws per minibatch: some uniform always lies within 1e-6 of its probability, where two correct fp32
    # implementations toss a coin -- the case's uniforms are moved 1e-4 away from p on the side they were on (same bits)
    name, meta = "c3_b512", _meta(fl_kw, 30, 512, 2)
    common.separate_draws(name, meta)
    flips = []
    want = common.oracle_train_case(name, meta, flips=flips)
    assert common.sampling_margin(want, meta, name) > 5e-5
    got, eng = common.hip_train_case(name, meta)
    common.assert_parity(got, want, flips, eng, "config3-b512", skip=("y2.bias",))
    got, eng = common.hip_train_case(name, meta, fused=True)          # ... and the fused step on what training sees
    keep = ("losses", "n_steps", "hits", "logs", "outp", "dist", ".g.", ".p.", "gradnorm")
    pick = lambda d: {k: v for k, v in d.items() if any(t in k for t in keep)}
    common.assert_parity(pick(got), pick(want), flips, eng, "config3-b512-fused", skip=("y2.bias",))


C1 = dict(use_binary=True, fixed_exchange=False, max_exchange=10, learning_rate=1e-4, entropy_rec=0.01, entropy_sen=0.01,
          entropy_s=0.08, img_feat_dim=512, img_h_dim=256, rec_w_dim=32, sender_out_dim=32, rec_hidden=64, wv_dim=100,
          baseline_hid_dim=500, top_k_train=6)


@pytest.mark.parametrize("batch", [1, 10, 50])
def test_ragged_batches_fused_step_vs_oracle(batch):
    """Batch sizes that fill neither a 16-row MFMA tile nor a wave: the fused training step (live-row list, k_baselines3,
    compacted k_wgrad, role launches) against the oracle."""
    meta = _meta(dict(C1, batch_size=batch), 30, batch, 2)
    got, eng = common.hip_train_case(None, meta, fused=True)
    flips = []
    want = common.oracle_train_case(None, meta, flips=flips)
    # a sample stops computing after its own stop step in the fused path: per-step arrays differ in entries every loss
    # masks out (test_hip_parity.py compares those in run-all mode); what training sees must agree
    keep = ("losses", "n_steps", "hits", "logs", "outp", "dist", ".g.", ".p.", "gradnorm")
    got = {k: v for k, v in got.items() if any(t in k for t in keep)}
    want = {k: v for k, v in want.items() if any(t in k for t in keep)}
    common.assert_parity(got, want, flips, eng, "ragged%d" % batch, skip=("y2.bias",))


@pytest.mark.parametrize("n_classes", [5, 30, 32])
def test_register_resident_path_class_counts(n_classes):
    """The register-resident kernels cover every class count up to 32 at the BASELINE agent shape (D = 30 has its own
    instantiation, the others run the capacity-32 one): fused training step vs the oracle, and the launch names show that
    neither the tile path nor the per-sample generic kernels ran."""
    batch = 16
    meta = _meta(dict(C1, batch_size=batch, top_k_train=min(6, n_classes - 1)), n_classes, batch, 2)
    got, eng = common.hip_train_case(None, meta, fused=True)
    flips = []
    want = common.oracle_train_case(None, meta, flips=flips)
    keep = ("losses", "n_steps", "hits", "logs", "outp", "dist", ".g.", ".p.", "gradnorm")
    got = {k: v for k, v in got.items() if any(t in k for t in keep)}
    want = {k: v for k, v in want.items() if any(t in k for t in keep)}
    common.assert_parity(got, want, flips, eng, "fastD%d" % n_classes, skip=("y2.bias",))
    x, target, desc, _ = common.case_inputs(meta, 0)
    dev = eng.device
    eng.set_profiling(True)
    eng.train_step(torch.from_numpy(x).to(dev), torch.from_numpy(target).to(dev), torch.from_numpy(desc).to(dev), seed=1)
    torch.cuda.synchronize()
    names = [n for n, _ in eng.kernel_times()]
    eng.set_profiling(False)
    # (Adaptive binary steps of the small agents: the fused launch of kernels_game.h; MMG_NO_GAME=1: k_conversation + k_bwd_conv)
    assert ("k_game" in names or "k_conversation" in names) and "k_conv_tile" not in names and "k_bwd_tile" not in names, names


@pytest.mark.parametrize("n_classes,batch", [(200, 40), (33, 16), (1000, 24)])
def test_many_class_binary_adaptive_vs_oracle(n_classes, batch):
    """Binary messages, Adaptive, more classes than the register-resident kernels hold (D > 32): the conversation runs on
    k_conversation_mc (a workgroup per sample that also owns a class slice of its 16-sample tile; ragged last tile, class
    slices that end before / after D), every per-step array of exchange() against the oracle in run-all mode, then the
    fused training step (no early exit on this path) on what training sees."""
    meta = _meta(dict(C1, batch_size=batch), n_classes, batch, 2)
    got, eng = common.hip_train_case(None, meta)
    flips = []
    want = common.oracle_train_case(None, meta, flips=flips)
    common.assert_parity(got, want, flips, eng, "mcD%d" % n_classes, skip=("y2.bias",))
    got_f, eng = common.hip_train_case(None, meta, fused=True)
    keep = ("losses", "n_steps", "hits", "logs", "outp", "dist", ".g.", ".p.", "gradnorm")
    pick = lambda d: {k: v for k, v in d.items() if any(t in k for t in keep)}
    common.assert_parity(pick(got_f), pick(want), flips, eng, "mcD%d-fused" % n_classes, skip=("y2.bias",))
    x, target, desc, _ = common.case_inputs(meta, 0)
    dev = eng.device
    eng.set_profiling(True)
    eng.train_step(torch.from_numpy(x).to(dev), torch.from_numpy(target).to(dev), torch.from_numpy(desc).to(dev), seed=1)
    torch.cuda.synchronize()
    names = [n for n, _ in eng.kernel_times()]
    eng.set_profiling(False)
    assert "k_conversation_mc" in names, names


@pytest.mark.parametrize("optim", ["Adam", "SGD"])
def test_fused_step_with_adam_and_sgd_at_config1_shape_vs_oracle(optim):
    """k_wgrad<OPT>'s in-launch clip + optimizer at config 1's shape (k_game_fast + k_wgrad<true>, the
    two launches of the metric config) met the oracle for RMSprop only; Adam / SGD were covered at tiny dimensions on the phased
    path (g3_tiny_*).  Two fused minibatches with -optim_type Adam and SGD against the oracle's torch.optim.Adam / SGD
    (model.py:1127-1135): post-update parameters of the first step feed the second."""
    z, meta = common.load_golden("g2_adaptive_c1")
    meta = dict(meta, optim_type=optim, learning_rate=1e-3 if optim == "SGD" else 1e-4)
    got, eng = common.hip_train_case(None, meta, fused=True)
    names = _kernel_names(eng, meta)
    assert names == ["k_game", "k_wgrad"], names                           # the optimizer ran inside k_wgrad's launch
    flips = []
    want = common.oracle_train_case(None, meta, flips=flips)
    common.assert_parity(_pick(got), _pick(want), flips, eng, "config1-fused-" + optim, skip=("y2.bias",))
