"""k_conversation_mc3p (two sample tiles per workgroup, kernels_mc3p.h) against k_conversation_mc3 (MMG_NO_MC3P=1) on the same
Philox-sampled minibatches of BASELINE config 5's agents: same lane maps and arithmetic, so the tape and the updated parameters
must be IDENTICAL bit for bit.  usage: mc3p_ab.py [batch ...]      exit code 1 on a mismatch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
cfg = dict(bench.WORKLOADS["c5"][0])
ok_all = True
for B in [int(a) for a in sys.argv[1:]] or [512, 528, 2048]:
    feats, target, desc = bench.synthetic_dataset(max(3000, 2 * B), cfg["n_classes"], 512, 100)
    res = []
    for off in (False, True):
        if off: os.environ["MMG_NO_MC3P"] = "1"
        eng = Engine(batch=B, **cfg)
        os.environ.pop("MMG_NO_MC3P", None)
        eng.load_state_dicts(init_state_dicts(eng, seed=0))
        dev = eng.device
        d = torch.from_numpy(desc).to(dev)
        eng.set_profiling(True)
        for i in range(2):
            x = torch.from_numpy(feats[B * i:B * i + B]).to(dev); t = torch.from_numpy(target[B * i:B * i + B]).to(dev)
            eng.train_step(x, t, d, seed=7)
        torch.cuda.synchronize()
        eng.check_sync()
        times = {}
        for n, ms in eng.kernel_times():
            times[n] = times.get(n, 0.0) + ms * 500.0       # us per minibatch (2 minibatches)
        eng.set_profiling(False)
        x = torch.from_numpy(feats[:B]).to(dev); t = torch.from_numpy(target[:B]).to(dev)
        for i in range(5): eng.train_step(x, t, d, seed=7)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for i in range(30): eng.train_step(x, t, d, seed=7)
        torch.cuda.synchronize()
        times["minibatch"] = 1e6 * (time.perf_counter() - t0) / 30
        res.append((eng.flat_params.clone(), {k: eng.tape[k].clone() for k in ("tstar", "logs", "dist", "hit", "gru", "h", "z", "s", "ps", "mask", "Astar", "hstar", "losses", "outp")}, times))
        del eng
    same = torch.equal(res[0][0], res[1][0]) and all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])
    bad = [k for k in res[0][1] if not torch.equal(res[0][1][k], res[1][1][k])]
    ok_all = ok_all and same
    print("B = %4d: pair kernel == single-tile kernel: %s %s | training minibatch %.1f us (pair) vs %.1f us (single), 30 steady-state steps" % (
        B, same, bad, res[0][2]["minibatch"], res[1][2]["minibatch"]))
print("OK" if ok_all else "MISMATCH")
sys.exit(0 if ok_all else 1)
