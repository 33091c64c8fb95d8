"""Timeline of k_game_fast (kernels_game.h) from the -DMMG_TIMING build: when every sample role passes its milestones, when the
baseline / statistics / class roles see the forward passes and finish -- the critical path of the fused launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalgame_amd import _lib
from multimodalgame_amd import build as _build
_lib.LIB_PATH = _build.build_timing_library()
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench
eng = Engine(batch=64, **dict(bench.C2))
eng.load_state_dicts(init_state_dicts(eng, 0))
feats, target, desc = bench.synthetic_dataset(3000, 30, 512, 100)
dev = eng.device
d = torch.from_numpy(desc).to(dev)
# PRETRAIN=n: n untimed minibatches first (cycling over 25 batches as bench.py's window does): the conversations of a trained pair
for it in range(int(os.environ.get("PRETRAIN", "0"))):
    k = it % 25
    eng.train_step(torch.from_numpy(feats[64 * k:64 * k + 64]).to(dev), torch.from_numpy(target[64 * k:64 * k + 64]).to(dev), d, seed=0)
for it in range(int(os.environ.get("ITERS", "12"))):
    x = torch.from_numpy(feats[64 * it:64 * it + 64]).to(dev); t = torch.from_numpy(target[64 * it:64 * it + 64]).to(dev)
    eng.tape["dbg2"].zero_()
    eng.train_step(x, t, d, seed=0)
torch.cuda.synchronize()
g = eng.tape["dbg2"].view(torch.int64).cpu().numpy().astype(np.float64)
us = lambda v: v * 10.0 / 1e3
S = g[2048:2048 + 16 * 64].reshape(64, 16)
t0 = S[:, 0].min()
names = ["start", "pairs held / loop start", "loop end", "pair A out (rows)", "epilogue 2 done", "output step done", "bases + first sweep", "second sweep", "stat pairs fresh", "coefficients", "scalars", "end", "t*", "pair B out (sums)"]
ts = S[:, 12].astype(int)
lon = int(np.argmax(S[:, 2]))
print("samples: t* min/mean/max %d / %.2f / %d; the longest conversation: sample %d (t* = %d)" % (ts.min(), ts.mean(), ts.max(), lon, ts[lon]))
print("%-26s %10s %10s %10s | sample %d" % ("milestone (us after start)", "min", "median", "max", lon))
for k, nm in [(0, names[0]), (14, "weights parked"), (1, names[1]), (2, names[2]), (3, names[3]), (13, names[13])] + [(k, names[k]) for k in range(4, 12)]:
    v = us(S[:, k] - t0)
    print("%-26s %10.2f %10.2f %10.2f | %8.2f" % (nm, v.min(), np.median(v), v.max(), v[lon]))
print("polls of the prep pairs per sample role: min %d median %d max %d" % (S[:, 15].min(), np.median(S[:, 15]), S[:, 15].max()))
pr = g[4608:4608 + 512].reshape(256, 2); pr = pr[pr[:, 0] > 0]
print("prep roles (%d): start %.2f..%.2f | done %.2f..%.2f; slowest %s" % ((len(pr),) + tuple(us(v - t0) for v in (pr[:, 0].min(), pr[:, 0].max(), pr[:, 1].min(), pr[:, 1].max())) + (np.argsort(-pr[:, 1])[:6].tolist(),)))
st = g[3072:3072 + 4 * 52].reshape(52, 4)
ok = st[:, 0] > 0
print("statistics waves: start %.2f..%.2f | forward passes seen %.2f..%.2f | done %.2f..%.2f" % tuple(us(v - t0) for v in (st[ok, 0].min(), st[ok, 0].max(), st[ok, 1].min(), st[ok, 1].max(), st[ok, 2].min(), st[ok, 2].max())))
nb = 0
while nb < 640 and g[3584 + 4 * nb] > 0: nb += 1
bs = g[3584:3584 + 4 * nb].reshape(nb, 4)
print("baseline roles (%d): start %.2f..%.2f | forward passes seen %.2f..%.2f | done %.2f..%.2f" % ((nb,) + tuple(us(v - t0) for v in (bs[:, 0].min(), bs[:, 0].max(), bs[:, 1].min(), bs[:, 1].max(), bs[:, 2].min(), bs[:, 2].max()))))
cl = g[3328:3328 + 60].reshape(30, 2)
print("class roles: released %.2f..%.2f | done %.2f..%.2f" % tuple(us(v - t0) for v in (cl[:, 0].min(), cl[:, 0].max(), cl[:, 1].min(), cl[:, 1].max())))
print("last sample end %.2f us; last done pair %.2f us" % (us(S[:, 11].max() - t0), us(S[:, 3].max() - t0)))
dur = us(bs[:, 2] - bs[:, 1])
order = np.argsort(-dur)[:12]
print("slowest baseline roles (role: which byi slot | seen -> done us):", [(int(r), int(r % 16 // 8), int(r % 8), int(r // 16), round(float(dur[r]), 2)) for r in order])
print("baseline role duration by slot:", [round(float(dur[np.arange(nb) // 16 == s].mean()), 2) for s in range(nb // 16)], "by which:", [round(float(dur[(np.arange(nb) % 16) // 8 == w].mean()), 2) for w in (0, 1)])
sd = us(st[ok, 2] - st[ok, 1])
print("statistics waves seen -> done: min %.2f median %.2f max %.2f; slowest waves %s" % (sd.min(), np.median(sd), sd.max(), np.argsort(-sd)[:8].tolist()))
print("live rows %d -> %d windows" % (int((ts + 1).sum()), (int((ts + 1).sum()) + 15) // 16))

bw = g[5120:5120 + 8 * nb].reshape(nb, 8)
okb = bw[:, 3] > 0
print("baseline role, first window (mean over %d roles): forward passes seen -> row list + barrier %.2f | -> row loads issued (+ basehx pairs held) %.2f | -> MFMAs done (the loads have arrived) %.2f | -> hidden stores + partial sums + barrier %.2f | -> role done %.2f" % (
    okb.sum(), us((bw[okb, 0] - bs[okb, 1]).mean()), us((bw[okb, 1] - bw[okb, 0]).mean()), us((bw[okb, 2] - bw[okb, 1]).mean()), us((bw[okb, 3] - bw[okb, 2]).mean()), us((bs[okb, 2] - bw[okb, 3]).mean())))
for w in (0, 1):
    m = okb & ((np.arange(nb) % 16) // 8 == w)
    print("   which = %d: %.2f | %.2f | %.2f | %.2f | %.2f" % (w, us((bw[m, 0] - bs[m, 1]).mean()), us((bw[m, 1] - bw[m, 0]).mean()), us((bw[m, 2] - bw[m, 1]).mean()), us((bw[m, 3] - bw[m, 2]).mean()), us((bs[m, 2] - bw[m, 3]).mean())))

# phases of sample 0's forward steps (tp.dbg[8 + 10 t + k], k = 0..6: the seven barrier-separated phases of kernels_fast3.h's step)
dbg = eng.tape["dbg"].view(torch.int64).cpu().numpy().astype(np.float64)
for t in range(10):
    st = dbg[8 + 10 * t:8 + 10 * t + 7]
    if st[0] == 0 or st[6] < st[0]: break
    prev = dbg[8 + 10 * (t - 1) + 6] if t > 0 else dbg[2]
    print("forward step %d of sample 0: %s | step total %.2f us" % (t, " ".join("%.2f" % us(b - a) for a, b in zip([prev] + list(st[:6]), st)), us(st[6] - prev)))
