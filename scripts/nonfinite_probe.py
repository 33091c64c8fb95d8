"""What a non-finite parameter does on the HIP path against the oracle (fmax_nn is v_max_f32, which returns the
OTHER operand for a NaN -- torch's relu propagates it).  One weight of receiver.y1 is set to NaN / +Inf / -Inf; prints which of the
six losses are finite on either side and whether the update was applied."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import common
from oracle import cpu_ref
z, meta = common.load_golden("g2_adaptive_c1")
meta = dict(meta, n_minibatches=1)
fl = common.flags_from_meta(meta)
for label, val in (("NaN", float("nan")), ("+Inf", float("inf")), ("-Inf", float("-inf"))):
    for key, pos in (("y1.weight", (3, 5)), ("w_h.weight", (2, 7)), ("y1.weight", (3, 64 + 10))):
        eng = common.make_engine(meta)
        eng.params["receiver"][key][pos] = val
        x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, 0, "g2_adaptive_c1")
        dev = eng.device
        a = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x, target, desc, u_z, u_s[..., 0], u_w)]
        before = eng.flat_params.clone()
        eng.train_step(*a)
        torch.cuda.synchronize()
        hip = list(eng.losses().values())[:6]
        moved = int((eng.flat_params != before).sum().item())
        nonfin = int((~torch.isfinite(eng.flat_params)).sum().item())
        # oracle
        torch.manual_seed(0)
        tape = cpu_ref.UniformTape()
        models = cpu_ref.build_agents(fl, rng=tape)
        cpu_ref.load_filled(models, seed=meta["seed_weights"])
        with torch.no_grad():
            dict(models["receiver"].named_parameters())[key][pos] = val
        opt = cpu_ref.build_optimizers(models, fl)
        tape.u = {"z": u_z, "s": u_s, "w": u_w}; tape.t = {"z": 0, "s": 0, "w": 0}
        res = cpu_ref.train_minibatch(models, opt, torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(desc), fl)
        ora = [float(res[k]) for k in ("nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen")]
        onf = sum(int((~torch.isfinite(p)).sum()) for m in models.values() for p in m.parameters())
        print("%-5s in receiver.%-10s%-9s HIP finite losses %s  non-finite parameters after the step %d (moved %d) | oracle finite losses %s  non-finite parameters %d" % (
            label, key, pos, [int(np.isfinite(v)) for v in hip], nonfin, moved, [int(np.isfinite(v)) for v in ora], onf))
        del eng
