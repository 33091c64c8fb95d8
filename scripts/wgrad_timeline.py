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
CASES = {"c2": dict(bench.C2), "c3": dict(bench.C2, fixed_exchange=True), "c3s": dict(bench.C2, fixed_exchange=True, batch=512),
         "c4": dict(bench.C2, w_dim=256, h_dim=1024), "c5s": dict(bench.C2, use_binary=False, fixed_exchange=True, n_classes=1000, batch=2048)}
cfg = CASES[os.environ.get("CASE", "c2")]            # CASE=c3s: config 3 with all 512 samples on one GPU, ...
NB = cfg.pop("batch", 64)
eng = Engine(batch=NB, **cfg)
eng.load_state_dicts(init_state_dicts(eng, 0))
feats, target, desc = bench.synthetic_dataset(max(3000, NB), cfg["n_classes"], 512, 100)
dev = eng.device
x = torch.from_numpy(feats[:NB]).to(dev); t = torch.from_numpy(target[:NB]).to(dev); d = torch.from_numpy(desc).to(dev)
for it in range(int(os.environ.get("PRETRAIN", "0"))):          # PRETRAIN=n: the conversations of a trained pair (bench.py's window)
    k = it % max(len(feats) // NB - 1, 1)
    eng.train_step(torch.from_numpy(feats[NB * k:NB * k + NB]).to(dev), torch.from_numpy(target[NB * k:NB * k + NB]).to(dev), d, seed=0)
for it in range(6):
    eng.tape["dbg2"].zero_()
    eng.train_step(x, t, d, seed=0)
torch.cuda.synchronize()
raw = eng.tape["dbg2"].view(torch.int64).cpu().numpy()
ph = raw[8192:8192 + 8000].reshape(-1, 4).astype(np.float64)
u = lambda v: v * 0.01
okp = (ph[:, 0] > 0) & (ph[:, 3] > 0)
if okp.any():
    g0 = raw[:8192].reshape(-1, 2)
    g0 = g0[g0[:, 0] > 0][:, 0].min()
    u = lambda v: v * 0.01
    print("GEMM tiles (%d): job + row list held at %.2f us after the first block started | rows reduced +%.2f | sum of squares out +%.2f | coefficient held +%.2f (mean); last pair out at %.2f, first coefficient at %.2f" % (
        okp.sum(), u((ph[okp, 0] - g0).mean()), u((ph[okp, 1] - ph[okp, 0]).mean()), u((ph[okp, 2] - ph[okp, 1]).mean()), u((ph[okp, 3] - ph[okp, 2]).mean()),
        u(ph[okp, 2].max() - g0), u(ph[okp, 3].min() - g0)))
pub = ph[:, 2]
okb = pub > 0
if okb.any():
    first = pub[okb].min()
    order = np.argsort(-pub)[:16]
    print("blocks by the time their sum of squares went out (us after the first one): median %.2f, p90 %.2f, last %.2f; the latest: %s" % (
        u(np.median(pub[okb]) - first), u(np.percentile(pub[okb], 90) - first), u(pub[okb].max() - first), [(int(b), round(float(u(pub[b] - first)), 2)) for b in order]))
red = u(ph[:, 1] - ph[:, 0]); held = ph[:, 0] > 0
nt = int(np.nonzero(held & (ph[:, 1] > 0))[0].max()) + 1
print("rows-reduced time by tile range (us, mean):", " ".join("%d-%d:%.1f" % (lo, min(lo + 24, nt) - 1, red[lo:lo + 25][(ph[lo:lo + 25, 1] > 0)].mean()) for lo in range(0, nt, 25)))
print("job + row list held, by tile range (us after the first):", " ".join("%d:%.1f" % (lo, u(ph[lo:lo + 25, 0][ph[lo:lo + 25, 0] > 0].mean() - ph[held, 0].min())) for lo in range(0, nt, 50)))
st = raw[:8192].reshape(-1, 2)
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
# the trailing blocks (column sums, spare block) one by one when asked: TAIL=n
ntail = int(os.environ.get("TAIL", "0"))
if ntail:
    ids = np.nonzero(nz)[0]
    print("last %d blocks (id:dur):" % ntail, " ".join("%d:%.0f" % (ids[i], dur[i]) for i in range(len(ids) - ntail, len(ids))))
