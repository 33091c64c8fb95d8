"""Philox4x32-10 known-answer test (CPU) and parity of the production sampling path (GPU)."""
import numpy as np
import pytest
import torch

from tests import common, philox_ref


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds; ctr = 0, key = 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    # (our generator returns word 0 >> 8)
    u = philox_ref.philox_uniform(0, np.array([0], np.uint32), 0, 0)
    assert int(u[0] * 16777216) == 0x6627e8d5 >> 8
    # ctr = ffffffff x4 needs c3 = 0xffffffff which this generator never uses; check key/counter sensitivity instead
    a = philox_ref.philox_uniform(1234, np.arange(1000, dtype=np.uint32), 7, 1)
    b = philox_ref.philox_uniform(1234, np.arange(1000, dtype=np.uint32), 8, 1)
    assert 0.45 < a.mean() < 0.55 and (a != b).mean() > 0.99 and a.min() >= 0 and a.max() < 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g2_adaptive_c1", "g5_one_active"])
def test_in_kernel_sampling_equals_oracle_with_philox_uniforms(name):
    """No injected uniforms: the kernels draw with Philox(seed, minibatch counter).  The oracle fed with the numpy
    Philox numbers must then see the same bits, masks, losses and parameters (fast and generic kernels)."""
    z, meta = common.load_golden(name)
    fl = common.flags_from_meta(meta)
    seed = 0xC0FFEE1234
    eng = common.make_engine(meta)
    dev = eng.device
    from oracle import cpu_ref
    tape = cpu_ref.UniformTape()
    models = cpu_ref.build_agents(fl, rng=tape)
    cpu_ref.load_filled(models, seed=meta["seed_weights"])
    opts = cpu_ref.build_optimizers(models, fl)
    for i in range(meta["n_minibatches"]):
        x, target, desc, _ = common.case_inputs(meta, i, None)
        xd, td, dd = [torch.from_numpy(a).to(dev) for a in (x, target, desc)]
        eng.train_step(xd, td, dd, seed=seed)
        torch.cuda.synchronize()
        u_z, u_s, u_w = philox_ref.conversation_uniforms(seed, i + 1, fl.max_exchange, meta["batch"], fl.rec_w_dim)
        tape.u = {"z": u_z, "s": u_s, "w": u_w}; tape.t = {"z": 0, "s": 0, "w": 0}
        res = cpu_ref.train_minibatch(models, opts, torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(desc), fl)
        n = res["n_steps"]
        L = eng.losses()
        assert int(L["n_steps"]) == n
        tstar = eng.tape["tstar"].cpu().numpy()
        for t in range(n):          # early exit: a sample's tape is valid up to its own last step
            act = tstar >= t
            np.testing.assert_array_equal(eng.tape["z"][t].cpu().numpy()[act], res["sen_feats"][t].numpy()[act])
            np.testing.assert_array_equal(eng.tape["s"][t].cpu().numpy()[act], res["s_feats"][t].numpy()[act])
            actn = tstar > t
            np.testing.assert_array_equal(eng.tape["w"][t].cpu().numpy()[actn], res["rec_feats"][t].detach().numpy()[actn])
        for k in ("nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen"):
            np.testing.assert_allclose(L[k], float(res[k].detach()), atol=1e-4, rtol=1e-3, err_msg=k)
    for agent, d in eng.params.items():
        for k, v in d.items():
            if k == "y2.bias":
                continue
            a, b = v.cpu().numpy(), models[agent].state_dict()[k].numpy()
            bad = ~np.isclose(a, b, rtol=1e-3, atol=1e-4)
            assert bad.mean() <= 2e-3, "%s.%s: %d / %d entries differ" % (agent, k, bad.sum(), bad.size)
