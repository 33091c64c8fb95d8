"""Parity of the HIP path (through the C-ABI) with the CPU oracle and with the golden vectors
generated from the reference.  Tolerances: north_star asks for logits / losses within 1e-4 of the
reference CPU path; sampled bits, masks, step counts and top-k hits must match exactly."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from tests import common

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-4, 1e-3     # forward quantities: 1e-4 absolute, no relative term (common.compare_packed); gradients: 1e-4 + 1e-3 |v|


def _skip_keys(meta):
    # continuous mode: the reference runs the (unused) baselines forward, this path does not
    # y2.bias: see compare_packed (its exact gradient is zero; the optimizer amplifies rounding noise)
    return ("y2.bias",) if meta["use_binary"] else ("y2.bias", ".bs", ".br")


def _key(problem):
    return problem.split(" ")[0]


@pytest.mark.parametrize("name", common.TRAIN_CASES)
def test_train_case_vs_golden_and_oracle(name):
    """Forward quantities, sampled bits, losses: must match the golden vectors (the reference's own run) AND the CPU oracle run
    on this host, 1e-4 absolute.  Gradients / updated parameters: must match this host's oracle -- where a ReLU unit sits
    within RELU_EPS of its threshold and the GPU put it on the other side, the oracle is re-run with that unit forced to the
    GPU's side and must then agree everywhere (common.assert_parity; nothing is excused).  Against the golden vectors (made on
    another host, whose oracle run may have put such a unit on either side: observed 6.9e-3 in y1.bias.grad between a Xeon
    and an EPYC host on g3_continuous mb1, one unit, dy = -1/B) a gradient entry may differ only where this host's oracle
    differs from the golden vectors too, or where the forced re-run was needed."""
    z, meta = common.load_golden(name)
    got, eng = common.hip_train_case(name, meta)
    flips = []
    want = common.oracle_train_case(name, meta, flips=flips)
    first = common.assert_parity(got, want, flips, eng, name + "/oracle", skip=_skip_keys(meta), atol=ATOL, rtol=RTOL)
    pg = common.compare_packed(got, z, atol=ATOL, rtol=RTOL, skip=_skip_keys(meta), shift_invariant=True, label=name + "/golden")
    hard = [p for p in pg if not common.is_grad_key(_key(p))]
    assert not hard, "forward mismatch vs golden (atol 1e-4, rtol 0):\n" + "\n".join(hard[:25])
    og = common.compare_packed(want, z, atol=ATOL, rtol=RTOL, skip=_skip_keys(meta), shift_invariant=True)
    explained = set(map(_key, og)) | set(map(_key, first))
    stray = [p for p in pg if _key(p) not in explained]
    assert not stray, "gradient entries that match this host's oracle but not the golden vectors, although the oracle does:\n" + "\n".join(stray[:25])


@pytest.mark.parametrize("name", ["g2_adaptive_c1", "g5_one_active", "g3_tiny_adam"])
def test_early_exit_and_fused_step_equal_run_all(name):
    """A sample that stops computing after its own stop step (training mode) must give the same
    losses, gradients and updated parameters as running every sample for all steps."""
    z, meta = common.load_golden(name)
    full, _ = common.hip_train_case(name, meta)
    for kw in (dict(early_exit=True), dict(fused=True)):
        got, _ = common.hip_train_case(name, meta, **kw)
        keys = [k for k in full if (".g." in k or ".p." in k or k.endswith("losses") or "gradnorm" in k)
                and "y2.bias" not in k]          # y2.bias: see common.compare_packed
        for k in keys:
            np.testing.assert_allclose(got[k], full[k], rtol=2e-4, atol=2e-6, err_msg="%s %s" % (kw, k))


@pytest.mark.parametrize("switch", ["MMG_NO_MERGE", "MMG_NO_FAST", "MMG_TILE"])
def test_kernel_variants_agree(switch, monkeypatch):
    """The same minibatches through (a) the default path (register-resident kernels; statistics / class reduction /
    basehx as workgroup roles of neighbouring launches), (b) MMG_NO_MERGE=1: those as their own launches,
    (c) MMG_NO_FAST=1: the generic any-shape kernels.  The switches are read at mmg_create."""
    name = "g2_adaptive_c1"
    z, meta = common.load_golden(name)
    want, _ = common.hip_train_case(name, meta, fused=True)
    monkeypatch.setenv(switch, "1")
    got, _ = common.hip_train_case(name, meta, fused=True)
    for k in want:
        if "y2.bias" in k or (not k.startswith("mb0.") and k.endswith((".y", ".outp"))):
            continue                                 # y2.bias and the per-row logit shift it causes: common.compare_packed
        if want[k].dtype.kind in "iub":
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        elif want[k].dtype.kind == "f":
            np.testing.assert_allclose(got[k], want[k], rtol=3e-4, atol=3e-6, err_msg=k)


def test_eval_pass_vs_golden():
    z, meta = common.load_golden("g4_eval_c1")
    fl = common.flags_from_meta(meta)
    eng = common.make_engine(meta)
    eng.params["receiver"]["s.bias"].fill_(1.2)
    x, target, desc = cpu_ref.synthetic_batch(meta["batch"], meta["n_classes"], 512, 100, seed=meta["seed_data"])
    dev = eng.device
    eng.forward(torch.from_numpy(x).to(dev), torch.from_numpy(target).to(dev), torch.from_numpy(desc).to(dev),
                train=False, run_all=True)
    torch.cuda.synchronize()
    n = int(z["n_steps"])
    tp = {k: v.cpu().numpy() for k, v in eng.tape.items() if k in ("mask", "s", "z", "w", "y", "dist", "tstar", "hit")}
    # the global break step: first t at which every sample's mask is zero
    alive = tp["mask"][1:, :, 0].sum(1)
    assert int(np.argmax(alive == 0)) + 1 == n
    np.testing.assert_array_equal(tp["s"][:n], z["s_feats"])
    np.testing.assert_array_equal(tp["z"][:n], z["sen_feats"])
    np.testing.assert_array_equal(tp["w"][:n], z["rec_feats"])
    np.testing.assert_array_equal(tp["mask"][:n], z["s_masks"][:n])
    np.testing.assert_allclose(tp["y"][:n], z["y"], atol=ATOL)
    np.testing.assert_allclose(tp["dist"], z["dist"], atol=ATOL)
    assert int(tp["hit"].sum()) == int(z["hits"])
    # "identical top-k index sets (ties: compare as sets)".  Per SAMPLE, not as a sum: the hit vector the kernel
    # stores against membership of the target in the reference's own top_k_ind (model.py:657-668), and the top-6 SETS of the
    # kernel's log-probabilities against the reference's wherever the 6th and 7th largest differ by more than the forward
    # tolerance (closer than that, two correct fp32 implementations may order them either way)
    k = z["top_k_ind"].shape[1]
    want_hit = (z["top_k_ind"] == target.reshape(-1, 1)).any(1)
    np.testing.assert_array_equal(tp["hit"].reshape(-1) != 0, want_hit)
    order = np.argsort(-z["dist"], axis=1, kind="stable")
    gap = np.take_along_axis(z["dist"], order[:, k - 1:k], 1) - np.take_along_axis(z["dist"], order[:, k:k + 1], 1)
    clear = gap.reshape(-1) > ATOL
    assert clear.sum() >= 0.9 * len(clear)
    got_top = np.argsort(-tp["dist"], axis=1, kind="stable")[:, :k]
    for b in np.nonzero(clear)[0]:
        assert set(got_top[b].tolist()) == set(z["top_k_ind"][b].tolist()), (b, sorted(got_top[b]), sorted(z["top_k_ind"][b]))
    np.testing.assert_array_equal(tp["s"][:n, :, 0].sum(0), z["conversation_lengths"])


def test_agent_level_steps_vs_golden():
    z, meta = common.load_golden("g1_agents_tiny")
    eng = common.make_engine(meta)
    dev = eng.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
    x, desc = t(z["x"]), t(z["desc"])
    u_z, u_s, u_w = z["u_z"], z["u_s"], z["u_w"]
    z0, p0, hx = eng.sender_forward(x, None, 0, True, u_z=t(u_z[0]))
    w_in = t(z["sen.w_in"])
    z1, p1, _ = eng.sender_forward(x, w_in, 1, True, u_z=t(u_z[1]))
    ze, pe, _ = eng.sender_forward(x, w_in, 1, False)
    np.testing.assert_array_equal(z0.cpu().numpy(), z["sen.train.t0.z"])
    np.testing.assert_array_equal(z1.cpu().numpy(), z["sen.train.t1.z"])
    np.testing.assert_array_equal(ze.cpu().numpy(), z["sen.eval.t1.z"])
    np.testing.assert_allclose(p0.cpu().numpy(), z["sen.train.t0.p"], atol=1e-5)
    np.testing.assert_allclose(p1.cpu().numpy(), z["sen.train.t1.p"], atol=1e-5)
    np.testing.assert_allclose(hx.cpu().numpy(), z["sen.h_x"], atol=1e-5)
    for mode in ("train", "eval"):
        h_z = torch.zeros(4, 5, device=dev)
        sprod = torch.ones(4, device=dev)
        for step, zin in enumerate((z0, z1)):
            s, sp, w, wp, y, h_w = eng.receiver_forward(zin, desc, h_z, sprod, step == 0, step, mode == "train",
                                                        u_s=t(u_s[step, :, 0]), u_w=t(u_w[step]))
            pre = "rec.%s.t%d." % (mode, step)
            np.testing.assert_array_equal(s.cpu().numpy(), z[pre + "s"])
            np.testing.assert_array_equal(w.cpu().numpy(), z[pre + "w"])
            np.testing.assert_allclose(sp.cpu().numpy(), z[pre + "s_prob"], atol=1e-5)
            np.testing.assert_allclose(wp.cpu().numpy(), z[pre + "w_prob"], atol=1e-5)
            np.testing.assert_allclose(y.cpu().numpy(), z[pre + "y"], atol=1e-5)
            np.testing.assert_allclose(h_z.cpu().numpy(), z[pre + "h_z"], atol=1e-5)
            np.testing.assert_allclose(h_w.cpu().numpy(), z[pre + "h_w"], atol=1e-5)
    bs = eng.baseline_forward("baseline_sen", hx, w_in, None)
    br = eng.baseline_forward("baseline_rec", None, z1, h_z)
    np.testing.assert_allclose(bs.cpu().numpy(), z["bas_sen"], atol=1e-5)
    np.testing.assert_allclose(br.cpu().numpy(), z["bas_rec"], atol=1e-5)


def test_graft_entry_smoke():
    """The driver's smoke() entry point (one fused minibatch checked against oracle and golden vectors)."""
    import __graft_entry__
    __graft_entry__.smoke()


def test_live_row_list_matches_tstar():
    """tape.rmap / rcount (device_utils: build_row_map, consumed by k_wgrad) = the (t, b) rows with t <= t*(b) in
    (t, b) order; tape.totals[3] counts them across steps."""
    name = "g2_adaptive_c1"
    z, meta = common.load_golden(name)
    _, eng = common.hip_train_case(name, meta, fused=True)
    torch.cuda.synchronize()
    tstar = eng.tape["tstar"].cpu().numpy()
    B, T = meta["batch"], meta["max_exchange"]
    want = [t * B + b for t in range(T) for b in range(B) if t <= tstar[b]]
    n = int(eng.tape["rcount"][0].item())
    assert n == len(want) == int((tstar + 1).sum())
    np.testing.assert_array_equal(eng.tape["rmap"][:n].cpu().numpy(), np.asarray(want, dtype=np.int32))
    assert float(eng.tape["totals"][3].item()) >= n          # running sum over the case's minibatches
