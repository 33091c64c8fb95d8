"""Pins the CPU oracle (oracle/cpu_ref.py) to the golden vectors produced by the
reference's own code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from tests import common


@pytest.mark.parametrize("name", common.TRAIN_CASES)
def test_train_case_matches_reference(name):
    z, meta = common.load_golden(name)
    got = common.oracle_train_case(name, meta)
    rec_is_binary = meta["use_binary"]
    exact_extra = () if rec_is_binary else ("rec_feats",)
    problems = common.compare_packed(got, z, atol=2e-6, rtol=2e-5)
    assert not problems, "\n".join(problems[:20])


def test_agents_tiny_match_reference():
    z, meta = common.load_golden("g1_agents_tiny")
    fl = common.flags_from_meta(meta)
    tape = cpu_ref.UniformTape()
    models = cpu_ref.build_agents(fl, rng=tape)
    for a, m in models.items():
        m.load_state_dict({k: torch.from_numpy(z["w.%s.%s" % (a, k)]) for k in m.state_dict().keys()})
    # the stored weights are what the repo-owned filler regenerates from the seed
    filled = cpu_ref.fill_state_dicts({a: {k: v.shape for k, v in m.state_dict().items()} for a, m in models.items()},
                                      seed=meta["seed_weights"])
    for a in filled:
        for k, v in filled[a].items():
            assert np.array_equal(v, z["w.%s.%s" % (a, k)])
    x, desc = torch.from_numpy(z["x"]), torch.from_numpy(z["desc"])
    u_z, u_s, u_w = z["u_z"], z["u_s"], z["u_w"]
    s, r = models["sender"], models["receiver"]
    s.train()
    tape.u["z"] = u_z; tape.t["z"] = 0
    z0, p0 = s(x, torch.zeros(4, 6), None, 0)
    w_in = torch.from_numpy(z["sen.w_in"])
    z1, p1 = s(x, w_in, None, 1)
    np.testing.assert_array_equal(z0.numpy(), z["sen.train.t0.z"])
    np.testing.assert_array_equal(z1.numpy(), z["sen.train.t1.z"])
    np.testing.assert_allclose(p0.detach().numpy(), z["sen.train.t0.p"], atol=1e-6)
    np.testing.assert_allclose(p1.detach().numpy(), z["sen.train.t1.p"], atol=1e-6)
    np.testing.assert_allclose(s.h_x.detach().numpy(), z["sen.h_x"], atol=1e-6)
    s.eval()
    ze, pe = s(x, w_in, None, 1)
    np.testing.assert_array_equal(ze.numpy(), z["sen.eval.t1.z"])
    for mode in ("train", "eval"):
        r.train() if mode == "train" else r.eval()
        r.reset_state()
        tape.u.update(s=u_s, w=u_w); tape.t.update(s=0, w=0)
        for t, zin in enumerate((z0, z1)):
            (sb, sp), (wf, wp), y = r(zin, desc)
            pre = "rec.%s.t%d." % (mode, t)
            np.testing.assert_array_equal(sb.numpy(), z[pre + "s"])
            np.testing.assert_array_equal(wf.detach().numpy(), z[pre + "w"])
            np.testing.assert_allclose(sp.detach().numpy(), z[pre + "s_prob"], atol=1e-6)
            np.testing.assert_allclose(wp.detach().numpy(), z[pre + "w_prob"], atol=1e-6)
            np.testing.assert_allclose(y.detach().numpy(), z[pre + "y"], atol=1e-6)
            np.testing.assert_allclose(r.h_z.detach().numpy(), z[pre + "h_z"], atol=1e-6)
            np.testing.assert_allclose(r.h_w.detach().numpy(), z[pre + "h_w"], atol=1e-6)
    np.testing.assert_allclose(models["baseline_sen"](s.h_x.detach(), w_in, None).detach().numpy(), z["bas_sen"], atol=1e-6)
    np.testing.assert_allclose(models["baseline_rec"](None, z1, r.h_z.detach()).detach().numpy(), z["bas_rec"], atol=1e-6)


def test_eval_pass_matches_reference():
    z, meta = common.load_golden("g4_eval_c1")
    fl = common.flags_from_meta(meta)
    models = cpu_ref.build_agents(fl)
    cpu_ref.load_filled(models, seed=meta["seed_weights"])
    with torch.no_grad():
        models["receiver"].s.bias.fill_(1.2)
    x, target, desc = cpu_ref.synthetic_batch(meta["batch"], meta["n_classes"], 512, 100, seed=meta["seed_data"])
    res = cpu_ref.eval_batch(models, torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(desc), fl)
    assert res["n_steps"] == int(z["n_steps"])
    assert 1 < res["n_steps"] < fl.max_exchange, "fixture should exercise the early break"
    np.testing.assert_array_equal(torch.stack(res["s_masks"]).numpy(), z["s_masks"])
    np.testing.assert_array_equal(torch.stack(res["s_feats"]).numpy(), z["s_feats"])
    np.testing.assert_array_equal(torch.stack(res["sen_feats"]).numpy(), z["sen_feats"])
    np.testing.assert_array_equal(torch.stack(res["rec_feats"]).numpy(), z["rec_feats"])
    np.testing.assert_allclose(torch.stack(res["y"]).numpy(), z["y"], atol=2e-6)
    np.testing.assert_allclose(res["dist"].numpy(), z["dist"], atol=2e-6)
    for a, b in zip(res["top_k_ind"].numpy(), z["top_k_ind"]):
        assert set(a.tolist()) == set(b.tolist())
    assert res["hits"] == int(z["hits"])
    np.testing.assert_array_equal(np.asarray(res["conversation_lengths"], np.float32), z["conversation_lengths"])


def test_golden_cases_exercise_what_they_claim():
    z, _ = common.load_golden("g2_adaptive_c1")
    n = int(z["mb0.n_steps"])
    masks = z["mb0.s_masks"][:, :, 0]
    assert masks.shape[0] == n + 1 and masks[0].all() and not masks[-1].any()
    assert 0 < masks[1].sum() < masks.shape[1], "stop bits should be mixed at step 0"
    z, _ = common.load_golden("g5_one_active")
    assert int(z["mb0.s_masks"][1].sum()) == 1
    z, _ = common.load_golden("g5_all_stop_first")
    assert int(z["mb0.n_steps"]) == 1 and float(z["mb0.losses"][2]) == 0.0
    z, meta = common.load_golden("g5_never_stop")
    assert int(z["mb0.n_steps"]) == meta["max_exchange"]
