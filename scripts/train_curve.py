#!/usr/bin/env python
"""Learning curve of the fused GPU training step on the learnable synthetic task of tests/test_hip_accuracy.py
(no oracle involved): train top-6 hit rate per 100 minibatches, then dev top-6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench

D, F, V, B = 30, 512, 100, 64
n_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
rs = np.random.RandomState(77)
proto = rs.standard_normal((D, F)).astype(np.float32)
desc = (0.3 * rs.standard_normal((D, V))).astype(np.float32)
def draw(n):
    t = rs.randint(0, D, size=(n,)).astype(np.int64)
    return np.abs(proto[t] + 0.3 * rs.standard_normal((n, F))).astype(np.float32), t
cfg = dict(bench.C2, learning_rate=1e-3)
eng = Engine(batch=B, **cfg)
eng.load_state_dicts(init_state_dicts(eng, seed=0))
dev = eng.device
dd = torch.from_numpy(desc).to(dev)
hits = []
for i in range(n_mb):
    x, t = draw(B)
    xd, td = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    if mode == "fused":
        eng.train_step(xd, td, dd, seed=5)
    else:
        eng.forward(xd, td, dd, seed=5, train=True, run_all=(mode == "runall"))
        eng.loss_stats(); eng.backward(xd, td, dd); eng.clip_step()
    hits.append(eng.tape["hit"].sum().item())
    if (i + 1) % 200 == 0:
        print("mb %5d  train top-6 %.1f %%  losses %s" % (i + 1, 100 * np.mean(hits[-200:]) / B, {k: round(v, 3) for k, v in eng.losses().items()}), flush=True)
xdev, tdev = draw(3008)
h = 0
for i in range(0, 3008, B):
    eng.forward(torch.from_numpy(xdev[i:i + B]).to(dev), torch.from_numpy(tdev[i:i + B]).to(dev), dd, train=False, run_all=True)
    h += int(eng.tape["hit"].sum().item())
print("dev top-6 %.2f %%" % (100.0 * h / 3008))
