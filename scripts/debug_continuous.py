"""One-off investigation (not collected by pytest): which gradient entries of g3_continuous differ between the golden
vectors (generated on the build host) and the oracle run on the GPU box -- a ReLU-mask flip of one |pre| ~ 1e-6 unit.
Run by hand on a GPU box: python scripts/debug_continuous.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import common
from oracle import cpu_ref
name = "g3_continuous"
z, meta = common.load_golden(name)
fl = common.flags_from_meta(meta)
# oracle step-by-step
tape = cpu_ref.UniformTape()
models = cpu_ref.build_agents(fl, rng=tape)
cpu_ref.load_filled(models, seed=meta["seed_weights"])
opts = cpu_ref.build_optimizers(models, fl)
eng = common.make_engine(meta)
dev = eng.device
for i in range(2):
    x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, i, name)
    tape.u = {"z": u_z, "s": u_s, "w": u_w}; tape.t = {"z": 0, "s": 0, "w": 0}
    res = cpu_ref.train_minibatch(models, opts, torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(desc), fl)
    xd, td, dd = torch.from_numpy(x).to(dev), torch.from_numpy(target).to(dev), torch.from_numpy(desc).to(dev)
    us = torch.from_numpy(np.ascontiguousarray(u_s[..., 0])).to(dev)
    eng.forward(xd, td, dd, None, us, None, train=True, run_all=True)
    eng.loss_stats(); eng.backward(xd, td, dd)
    torch.cuda.synchronize()
    print("== minibatch", i)
    yh = eng.tape["y"].cpu().numpy(); yo = np.stack([t.detach().numpy() for t in res["y"]])
    print("y maxdiff", np.abs(yh - yo).max(), "y range", np.abs(yo).max())
    zh = eng.tape["z"].cpu().numpy(); zo = np.stack([t.detach().numpy() for t in res["sen_feats"]])
    print("z maxdiff", np.abs(zh - zo).max(), np.abs(zo).max())
    for k, g in res["grads"]["receiver"].items():
        gh = eng.grads["receiver"][k].cpu().numpy(); go = g.numpy()
        d = np.abs(gh - go)
        print("  grad %-16s max|g| %.3e maxdiff %.3e  n(|g|<1e-7)=%d/%d" % (k, np.abs(go).max(), d.max(), (np.abs(go) < 1e-7).sum(), go.size))
    eng.clip_step(); torch.cuda.synchronize()
    for k, p in models["receiver"].state_dict().items():
        ph = eng.params["receiver"][k].cpu().numpy(); d = np.abs(ph - p.numpy())
        print("  param %-16s maxdiff %.3e  n(diff>5e-4)=%d/%d" % (k, d.max(), (d > 5e-4).sum(), d.size))
print("==== packed compare on this box")
got_o = common.oracle_train_case(name, meta)
for k in z.files:
    if k.startswith("mb1.g.") or k.startswith("mb1.losses") or k.startswith("mb1.y") or k.startswith("mb1.sen_feats") or k.startswith("mb1.rec_feats"):
        a, b = np.asarray(got_o[k], dtype=np.float64), np.asarray(z[k], dtype=np.float64)
        print("oracle-vs-golden %-40s %.3e" % (k, np.abs(a - b).max()))
print(torch.__version__, torch.get_num_threads(), torch.backends.mkldnn.is_available())
