"""Per-phase timeline of k_conversation_mc for workgroup 0 (tile 0, member 0) -- needs the -DMMG_TIMING build (compiled on
demand).  usage: mc_timeline.py [batch]   (config 5's agents: D = 1000, continuous, Fixed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalgame_amd import _lib
from multimodalgame_amd import build as _build
_lib.LIB_PATH = _build.build_timing_library()
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench
cfg, B, _ = bench.WORKLOADS["c5"]
cfg = dict(cfg)
if len(sys.argv) > 1:
    B = int(sys.argv[1])
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
print("weights issued %.2f | state init %.2f | loop %.2f us" % (us(0, 1), us(1, 2), us(2, 3)))
names = ["sample phases 1-5", "publish A + wait", "gather A", "logits+softmax+mixture", "publish partials + wait", "gather partials", "combine, g, w"]
tot = np.zeros(7)
T = cfg["max_exchange"]
for st in range(T):
    base = 16 + 8 * st
    prev = 2 if st == 0 else base - 8 + 6
    seg = [us(prev, base)] + [us(base + k, base + k + 1) for k in range(6)]
    tot += np.array(seg)
    print("step %d: %s = %.2f" % (st, " | ".join("%.2f" % v for v in seg), sum(seg)))
print("mean per step: " + " | ".join("%s %.2f" % (n, v / T) for n, v in zip(names, tot)) + " = %.2f us" % (tot.sum() / T))
if dbg[16 + 7]:
    c1 = np.mean([us(16 + 8 * st + 2, 16 + 8 * st + 7) for st in range(T)]); c2 = np.mean([us(16 + 8 * st + 7, 200 + st) for st in range(T)])
    c3 = np.mean([us(200 + st, 16 + 8 * st + 3) for st in range(T)])
    print("   of logits+softmax+mixture: class logits %.2f | tape / selected logits / softmax numerators %.2f | mixture MFMA %.2f" % (c1, c2, c3))
