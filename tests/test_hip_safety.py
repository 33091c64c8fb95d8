"""A timed-out in-launch dependency wait (device_utils.h: role_wait -> sync[511]) must never train on silently -- and must not
end the run either (round 6, fail-soft): k_opt / the norm role skip the update of that minibatch, the next call that starts a
minibatch clears the error, re-selects the launches WITHOUT in-launch waits, warns once and training continues.  The reference
has no such failure mode (model.py:1218-1330 simply keeps training)."""
import numpy as np
import pytest
import torch

from multimodalgame_amd import _lib
from tests import common

pytestmark = pytest.mark.gpu


def _inputs(meta, eng):
    x, target, desc, _ = common.case_inputs(meta, 0)
    return [torch.from_numpy(a).to(eng.device) for a in (x, target, desc)]


def test_dependency_timeout_skips_one_update_then_continues_on_the_launches_without_waits(monkeypatch):
    """force a timeout (sync[511] = 2, what role_wait stores when dependency 1 expires); observe exactly ONE
    skipped update; the next step warns (MmgWarning, return code 1 of the C call), reports mmg_degraded() == 2 and UPDATES; from
    there on the engine must take, bit for bit, the steps of an engine created on the fallback path (MMG_NO_ROLES=1) from the
    same parameters, optimizer state and sampling counters."""
    z, meta = common.load_golden("g2_adaptive_c1")
    eng = common.make_engine(meta)
    xd, td, dd = _inputs(meta, eng)
    assert eng.degraded() == 0
    eng.set_profiling(True)
    eng.train_step(xd, td, dd, seed=3)                       # a healthy step on the role launches
    torch.cuda.synchronize()
    assert [n for n, _ in eng.kernel_times()] == ["k_game", "k_wgrad"]
    eng.set_profiling(False)
    before, state_before = eng.flat_params.clone(), eng.opt_state.clone()
    eng.tape["sync"][511] = 2
    eng.train_step(xd, td, dd, seed=3)                       # the error word is set: the norm role must leave everything untouched
    torch.cuda.synchronize()
    torch.testing.assert_close(eng.flat_params, before, rtol=0, atol=0)
    torch.testing.assert_close(eng.opt_state, state_before, rtol=0, atol=0)
    with pytest.warns(_lib.MmgWarning, match="timed out.*continues on the launches without in-launch waits"):
        eng.train_step(xd, td, dd, seed=3)                   # recovers: clears the words, re-selects, trains
    torch.cuda.synchronize()
    assert eng.degraded() == 2
    eng.check_sync()                                         # the device word is clear again
    assert not torch.equal(eng.flat_params, before)          # ... and this minibatch DID update
    monkeypatch.setenv("MMG_NO_ROLES", "1")
    ref = common.make_engine(meta)                           # the fallback path from mmg_create on
    monkeypatch.delenv("MMG_NO_ROLES")
    assert ref.degraded() == 1
    ref.flat_params.copy_(eng.flat_params); ref.opt_state.copy_(eng.opt_state); ref.tape["counter"].copy_(eng.tape["counter"])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", _lib.MmgWarning)      # no further warning: one event, one line
        eng.set_profiling(True)
        for _ in range(4):
            eng.train_step(xd, td, dd, seed=3)
            ref.train_step(xd, td, dd, seed=3)
    torch.cuda.synchronize()
    names = [n for n, _ in eng.kernel_times()]
    assert "k_game" not in names and "k_opt" in names and any(n.startswith("k_prep") for n in names), names
    assert torch.equal(eng.flat_params, ref.flat_params) and torch.equal(eng.opt_state, ref.opt_state)
    assert torch.equal(eng.tape["losses"], ref.tape["losses"])


def test_no_roles_path_matches_the_role_launches():
    """The launches without in-launch waits (MMG_NO_ROLES=1: k_prep | k_conversation_fast3 | k_baselines3 | k_stats |
    k_bwd_conv_fast | k_dC | k_wgrad | k_opt) take the same training steps as k_game_fast + k_wgrad<OPT> within rounding."""
    import os
    z, meta = common.load_golden("g2_adaptive_c1")
    eng = common.make_engine(meta)
    os.environ["MMG_NO_ROLES"] = "1"
    try:
        ref = common.make_engine(meta)
    finally:
        del os.environ["MMG_NO_ROLES"]
    xd, td, dd = _inputs(meta, eng)
    for _ in range(3):
        eng.train_step(xd, td, dd, seed=7)
        ref.train_step(xd, td, dd, seed=7)
    torch.cuda.synchronize()
    eng.check_sync(); ref.check_sync()
    la, lb = eng.tape["losses"].cpu(), ref.tape["losses"].cpu()
    assert torch.equal(la[6:], lb[6:])                       # executed steps, hits
    torch.testing.assert_close(la[:6], lb[:6], rtol=2e-5, atol=2e-5)
    diff = (eng.flat_params - ref.flat_params).abs()
    assert float((diff > 2e-5).float().mean()) < 1e-4 and float(diff.max()) < 1e-2


def test_cu_budget_selects_launches_that_fit_it():
    """mmg_config.cu_budget: co-residency is sized from the compute units the CALLER can count on,
    not from the whole chip.  With 48 CUs k_game_fast's ~240 roles and k_wgrad<OPT>'s ~970 blocks cannot be co-resident: the
    launches that need no co-residency are selected at mmg_create (no timeout ever happens), and the step is the same step."""
    z, meta = common.load_golden("g2_adaptive_c1")
    eng = common.make_engine(meta)
    small = common.make_engine(meta, cu_budget=48)
    xd, td, dd = _inputs(meta, eng)
    small.set_profiling(True)
    for e in (eng, small):
        e.train_step(xd, td, dd, seed=5)
    torch.cuda.synchronize()
    names = [n for n, _ in small.kernel_times()]
    assert "k_game" not in names and "k_opt" in names, names
    small.check_sync()
    la, lb = eng.tape["losses"].cpu(), small.tape["losses"].cpu()
    assert torch.equal(la[6:], lb[6:])
    torch.testing.assert_close(la[:6], lb[:6], rtol=2e-5, atol=2e-5)


def _dp_engine(meta):
    """An engine configured as one rank of a 2-rank job (global_batch = 2 x batch): the phased entry points only."""
    eng = common.make_engine(meta, global_batch=2 * int(meta["batch"]))
    return eng, _inputs(meta, eng)


def _phased_step(eng, xd, td, dd):
    eng.forward(xd, td, dd, seed=3, train=True, run_all=False)
    eng.loss_stats()
    eng.backward(xd, td, dd)
    eng.clip_step()


def test_phased_path_skips_update_and_the_next_minibatch_recovers():
    """The data-parallel path (forward / loss_stats / backward / clip_step called separately): the flag goes out in the tail quad
    of the gradient buffer, k_opt leaves everything untouched, the calls in the MIDDLE of that minibatch do not fail, and the
    next mmg_exchange_forward(train) recovers."""
    z, meta = common.load_golden("g2_adaptive_c1")
    eng, (xd, td, dd) = _dp_engine(meta)
    _phased_step(eng, xd, td, dd)
    torch.cuda.synchronize()
    assert float(eng.flat_grads[-4]) == 0.0                  # healthy step: the flag quad behind the gradients is clear
    before = eng.flat_params.clone()
    eng.tape["sync"][511] = 2
    _phased_step(eng, xd, td, dd)                            # mmg_backward raises the flag, k_opt leaves everything untouched
    torch.cuda.synchronize()
    assert float(eng.flat_grads[-4]) == 1.0
    torch.testing.assert_close(eng.flat_params, before, rtol=0, atol=0)
    with pytest.warns(_lib.MmgWarning, match="timed out"):
        eng.forward(xd, td, dd, seed=3, train=True, run_all=False)
    eng.loss_stats(); eng.backward(xd, td, dd); eng.clip_step()
    torch.cuda.synchronize()
    assert eng.degraded() == 2 and float(eng.flat_grads[-4]) == 0.0
    assert not torch.equal(eng.flat_params, before)


def test_remote_rank_error_reaches_this_rank_through_the_gradient_all_reduce():
    """Another rank's flag arrives summed into the tail quad of the gradient buffer: this rank skips the update too, and all
    ranks switch to the launches without in-launch waits at their next minibatch (each sees a posted word: its own code or 1001)."""
    z, meta = common.load_golden("g2_adaptive_c1")
    eng, (xd, td, dd) = _dp_engine(meta)
    _phased_step(eng, xd, td, dd)
    torch.cuda.synchronize()
    before = eng.flat_params.clone()
    eng.forward(xd, td, dd, seed=3, train=True, run_all=False)
    eng.loss_stats()
    eng.backward(xd, td, dd)
    eng.flat_grads[-4] += 1.0                                # what the all-reduce (sum) would add from the failing rank
    eng.clip_step()
    torch.cuda.synchronize()
    torch.testing.assert_close(eng.flat_params, before, rtol=0, atol=0)
    with pytest.warns(_lib.MmgWarning, match="another rank"):
        eng.forward(xd, td, dd, seed=3, train=True)
    assert eng.degraded() == 2


def test_clear_error_keeps_the_selected_launches():
    """mmg_clear_error: the caller's own recovery (e.g. the other tenant of the GPU is gone) -- words cleared, role launches kept."""
    z, meta = common.load_golden("g2_adaptive_c1")
    eng = common.make_engine(meta)
    xd, td, dd = _inputs(meta, eng)
    eng.train_step(xd, td, dd, seed=3)
    eng.tape["sync"][511] = 2
    eng.train_step(xd, td, dd, seed=3)                       # skipped
    torch.cuda.synchronize()
    before = eng.flat_params.clone()
    eng.clear_error()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", _lib.MmgWarning)
        eng.set_profiling(True)
        eng.train_step(xd, td, dd, seed=3)
    torch.cuda.synchronize()
    assert eng.degraded() == 0 and [n for n, _ in eng.kernel_times()] == ["k_game", "k_wgrad"]
    assert not torch.equal(eng.flat_params, before)
    eng.check_sync()


def test_prep_roles_inside_the_conversation_launch_equal_k_prep_as_a_launch(monkeypatch):
    """mmg_exchange_forward runs k_prep's blocks as roles of k_conversation_fast3's launch when every role has a CU (kernels_fast3.h,
    MERGED; hand-off by (value, epoch) pairs, the minibatch counter bumped by the closing role).  MMG_NO_MERGE_PREP=1 (read by
    mmg_create) keeps k_prep a launch of its own: same constants bit for bit, same Philox streams (the counter arithmetic), the
    same training trajectory within rounding of the one value the consumers re-form themselves (cy)."""
    z, meta = common.load_golden("g2_adaptive_c1")
    x, target, desc, _ = common.case_inputs(meta, 0)
    eng_a = common.make_engine(meta)
    monkeypatch.setenv("MMG_NO_MERGE_PREP", "1")
    eng_b = common.make_engine(meta)
    monkeypatch.delenv("MMG_NO_MERGE_PREP")
    eng_b.flat_params.copy_(eng_a.flat_params); eng_b.opt_state.copy_(eng_a.opt_state)
    names = [[], []]
    first = [{}, {}]
    for k, eng in enumerate((eng_a, eng_b)):
        xd, td, dd = [torch.from_numpy(a).to(eng.device) for a in (x, target, desc)]
        for step in range(6):                                  # in-kernel Philox: the streams depend on the minibatch counter
            if step == 5:
                eng.set_profiling(True)
            eng.train_step(xd, td, dd, seed=11)
            if step == 0:                                      # same parameters: the constants and the draws must agree exactly
                torch.cuda.synchronize()
                first[k] = {key: eng.tape[key].clone() for key in ("hx", "Cd", "Dd", "hw0", "tstar", "z", "s")}
        torch.cuda.synchronize()
        names[k] = [n for n, _ in eng.kernel_times()]
        eng.set_profiling(False)
        eng.check_sync()
    assert not any(n.startswith("k_prep") for n in names[0]) and any(n.startswith("k_prep") for n in names[1]), names
    for key in first[0]:
        assert torch.equal(first[0][key], first[1][key]), key
    assert torch.equal(eng_a.tape["counter"], eng_b.tape["counter"]) and int(eng_a.tape["counter"][0]) == 6
    # five more updates: RMSprop turns a rounding-level difference of a near-zero gradient into a visible step of that element, so
    # the trajectories are compared in bulk -- all but a handful of the 384 k parameters within 2e-5, none further than 1e-2
    diff = (eng_a.flat_params - eng_b.flat_params).abs()
    assert float((diff > 2e-5).float().mean()) < 1e-4 and float(diff.max()) < 1e-2, (float((diff > 2e-5).float().mean()), float(diff.max()))


@pytest.mark.parametrize("merge_prep", [True, False])
def test_eval_forward_leaves_the_training_sampling_stream_alone(merge_prep, monkeypatch):
    """k_prep bumped the Philox minibatch counter on EVERY forward pass.  In a single-process run eval_dev shares the
    training engine whenever -batch_size_dev == -batch_size, so each evaluation moved the training sampling stream -- while a
    data-parallel job evaluates on rank 0 only, on a separate engine: the two runs then diverged.  An evaluation pass draws
    nothing and must leave counter[0] alone (the launch epoch counter[3] of the pair hand-offs still moves); training with an
    evaluation pass in between must take exactly the updates of training without it."""
    if not merge_prep:
        monkeypatch.setenv("MMG_NO_MERGE_PREP", "1")
    z, meta = common.load_golden("g2_adaptive_c1")
    x, target, desc, _ = common.case_inputs(meta, 0)
    eng_a, eng_b = common.make_engine(meta), common.make_engine(meta)
    eng_b.flat_params.copy_(eng_a.flat_params); eng_b.opt_state.copy_(eng_a.opt_state)
    xd, td, dd = [torch.from_numpy(a).to(eng_a.device) for a in (x, target, desc)]
    for step in range(4):
        eng_a.train_step(xd, td, dd, seed=5)
        eng_b.train_step(xd, td, dd, seed=5)
        if step in (0, 2):
            c0 = eng_b.tape["counter"].clone()
            eng_b.forward(xd, td, dd, seed=5, train=False, run_all=True)          # an eval_dev batch on the training engine
            torch.cuda.synchronize()
            c1 = eng_b.tape["counter"]
            assert int(c1[0]) == int(c0[0]) and int(c1[3]) == int(c0[3]) + 1, (c0.tolist(), c1.tolist())
    torch.cuda.synchronize()
    eng_a.check_sync(); eng_b.check_sync()
    assert int(eng_a.tape["counter"][0]) == int(eng_b.tape["counter"][0]) == 4
    assert torch.equal(eng_a.flat_params, eng_b.flat_params)


def test_specialised_paths_agree_with_their_fallbacks_under_early_stopping():
    """scripts/path_ab.py: every specialised kernel path against its fallbacks (down to the generic per-sample kernels) on the same
    Philox-sampled first minibatch with early stopping -- identical step / hit counts, losses within 2e-5 relative, and the default
    path reproducing itself bit for bit.  (Round 5: the check that would have caught k_conv_persist's sender roles running ahead of
    live samples in rounds 3-4; the oracle tests of that shape ran in run-all mode.)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "path_ab.py"), "2"], capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]


@pytest.mark.parametrize("fused_opt", [True, False])
@pytest.mark.parametrize("value,key,pos", [(float("inf"), "y1.weight", (3, 5)), (float("-inf"), "y1.weight", (3, 74)),
                                           (float("nan"), "y1.weight", (3, 74)), (float("nan"), "w_h.weight", (2, 7)),
                                           (float("nan"), "y1.weight", (3, 5)), (float("nan"), "rnn.weight_hh", (70, 3))])
def test_non_finite_parameter_shows_up_in_the_losses(value, key, pos, fused_opt, monkeypatch):
    """A non-finite parameter (a diverged run) must not train on silently.  The class-logit ReLU is v_max_f32 (device_utils.h:
    fmax_nn), which returns the OTHER operand for a NaN where torch's relu propagates it: a NaN in the h-part of receiver.y1.weight
    or in the GRU state reads as "unit off", and the NLL of that step would be a plausible log D while every loss of the reference is
    NaN (scripts/nonfinite_probe.py; ADVICE r04 asked for the behaviour to be pinned).  The backward pass carries the NaN, so the
    optimizer's norm stage (k_wgrad<OPT>'s norm role / k_opt) makes the logged NLL NaN when any agent's gradient norm is not finite:
    the NLL is non-finite whenever the oracle's is, in the SAME step, for +-Inf and NaN in and outside that ReLU's operand,
    and finite whenever all six losses of the oracle are."""
    from oracle import cpu_ref
    if not fused_opt:
        monkeypatch.setenv("MMG_NO_WGRAD_OPT", "1")
    z, meta = common.load_golden("g2_adaptive_c1")
    meta = dict(meta, n_minibatches=1)
    fl = common.flags_from_meta(meta)
    eng = common.make_engine(meta)
    eng.params["receiver"][key][pos] = value
    x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, 0, "g2_adaptive_c1")
    dev = eng.device
    a = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x, target, desc, u_z, u_s[..., 0], u_w)]
    eng.train_step(*a)
    hip_nll = float(list(eng.losses().values())[0])
    torch.manual_seed(0)
    tape = cpu_ref.UniformTape()
    models = cpu_ref.build_agents(fl, rng=tape)
    cpu_ref.load_filled(models, seed=meta["seed_weights"])
    with torch.no_grad():
        dict(models["receiver"].named_parameters())[key][pos] = value
    opt = cpu_ref.build_optimizers(models, fl)
    tape.u = {"z": u_z, "s": u_s, "w": u_w}; tape.t = {"z": 0, "s": 0, "w": 0}
    res = cpu_ref.train_minibatch(models, opt, torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(desc), fl)
    ora = [float(res[k].detach()) for k in ("nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen")]
    # never later than the reference ...
    assert np.isfinite(hip_nll) or not np.isfinite(ora[0]) or not all(np.isfinite(ora)), (hip_nll, ora)
    assert not np.isfinite(hip_nll) if not np.isfinite(ora[0]) else True, (hip_nll, ora)
    # ... and only when the reference's own step is broken (w_h feeds the message head only: the reference keeps a finite NLL for
    # this one step and shows the NaN in loss_binary_rec, then in every parameter of the receiver; the guard reports it one step early)
    if not np.isfinite(hip_nll):
        assert not all(np.isfinite(ora)), (hip_nll, ora)
