"""Per-block timing of k_wgrad (needs the -DMMG_TIMING build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalgame_amd import _lib
from multimodalgame_amd import build as _build
_lib.LIB_PATH = _build.build_timing_library()      # -DMMG_TIMING build, compiled on demand (never shipped with the tree)
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench
eng = Engine(batch=64, **bench.C2)
eng.load_state_dicts(init_state_dicts(eng, 0))
feats, target, desc = bench.synthetic_dataset(3000, 30, 512, 100)
dev = eng.device
x = torch.from_numpy(feats[:64]).to(dev); t = torch.from_numpy(target[:64]).to(dev); d = torch.from_numpy(desc).to(dev)
for it in range(6):
    eng.train_step(x, t, d, seed=0)
torch.cuda.synchronize()
st = eng.tape["dbg2"].view(torch.int64).cpu().numpy().reshape(-1, 2)
nz = st[:, 0] > 0
st = st[nz]
t0 = st[:, 0].min()
start = (st[:, 0] - t0) * 0.01; end = (st[:, 1] - t0) * 0.01      # us (100 MHz clock)
dur = end - start
print("blocks", len(st), "kernel span %.2f us" % end.max())
print("start: p50 %.2f p90 %.2f max %.2f" % (np.percentile(start, 50), np.percentile(start, 90), start.max()))
print("dur:   p50 %.2f p90 %.2f max %.2f" % (np.percentile(dur, 50), np.percentile(dur, 90), dur.max()))
order = np.argsort(-end)[:12]
for i in order:
    print("  block %4d start %.2f dur %.2f end %.2f" % (np.nonzero(nz)[0][i], start[i], dur[i], end[i]))
# by block-index ranges of 128
for lo in range(0, len(st), 128):
    sl = slice(lo, lo + 128)
    print("blocks %4d-%4d: start %.2f..%.2f  dur mean %.2f max %.2f  end max %.2f" % (lo, lo + 127, start[sl].min(), start[sl].max(), dur[sl].mean(), dur[sl].max(), end[sl].max()))
