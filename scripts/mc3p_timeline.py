"""Per-block timeline of k_conversation_mc3p for workgroup 0 (pair 0, member 0) -- needs the -DMMG_TIMING build (compiled on
demand).  usage: mc3p_timeline.py [batch]   (config 5's agents: D = 1000, continuous, Fixed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalgame_amd import _lib
from multimodalgame_amd import build as _build
_lib.LIB_PATH = _build.build_timing_library()
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench
cfg = dict(bench.WORKLOADS["c5"][0])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = Engine(batch=B, **cfg)
eng.load_state_dicts(init_state_dicts(eng, 0))
feats, target, desc = bench.synthetic_dataset(max(3000, B), cfg["n_classes"], 512, 100)
dev = eng.device
x = torch.from_numpy(feats[:B]).to(dev); t = torch.from_numpy(target[:B]).to(dev); d = torch.from_numpy(desc).to(dev)
for it in range(4):
    eng.train_step(x, t, d, seed=0)
torch.cuda.synchronize()
dbg = eng.tape["dbg"].view(torch.int64).cpu().numpy()
us = lambda a, b: (dbg[b] - dbg[a]) * 10.0 / 1e3          # s_memrealtime: 100 MHz
T = cfg["max_exchange"]
print("loop %.2f us (%.2f per step pair)" % (us(2, 3), us(2, 3) / T))
names = ["S(0,t)", "K(1,t-1)", "S(1,t)", "C(0,t)", "C(1,t)", "K(0,t)"]
tot = np.zeros(6); polls = np.zeros(4)
for st in range(T):
    b = 16 + 16 * st
    seg = [us(b + k, b + k + 1) for k in range(5)] + [us(b + 5, b + 16)]
    tot += np.array(seg)
    pw = [us(b + 3, b + 8), us(b + 4, b + 9), us(b + 5, b + 10), (us(b + 1, b + 11) if st > 0 else 0.0)]
    polls += np.array(pw)
    print("step %d: " % st + " | ".join("%s %.2f" % (n, v) for n, v in zip(names, seg)) + "   polls: A0 %.2f A1 %.2f P0 %.2f P1 %.2f" % tuple(pw))
print("mean: " + " | ".join("%s %.2f" % (n, v / T) for n, v in zip(names, tot)) + " = %.2f us per step pair" % (tot.sum() / T))
print("of which waiting in the polls (block start -> all pairs fresh): A(0) %.2f  A(1) %.2f  P(0) %.2f  P(1) %.2f" % tuple(polls / np.array([T, T, T, T - 1])))
