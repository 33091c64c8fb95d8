"""Per-phase timeline of k_conversation_fast2 / k_bwd_conv_fast / k_baselines2 for sample 0 (needs libmmg_timing.so: -DMMG_TIMING build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalgame_amd import _lib
from multimodalgame_amd import build as _build
_lib.LIB_PATH = _build.build_timing_library()      # -DMMG_TIMING build, compiled on demand (never shipped with the tree)
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench
cfg = dict(bench.C2); cfg["fixed_exchange"] = bool(int(os.environ.get("FIXED", "1")))
eng = Engine(batch=64, **cfg)
eng.load_state_dicts(init_state_dicts(eng, 0))
feats, target, desc = bench.synthetic_dataset(3000, 30, 512, 100)
dev = eng.device
x = torch.from_numpy(feats[:64]).to(dev); t = torch.from_numpy(target[:64]).to(dev); d = torch.from_numpy(desc).to(dev)
for it in range(6):
    if os.environ.get("FWD_ONLY"):       # forward launches only (k_prep, conversation, baselines): the conversation kernel's code stays in the I-cache
        eng.forward(x, t, d, seed=0, train=True, run_all=False, minimal=True)
    else:
        eng.train_step(x, t, d, seed=0)
torch.cuda.synchronize()
dbg = eng.tape["dbg"].view(torch.int64).cpu().numpy()
tick = 10.0  # wall_clock64: 100 MHz -> 10 ns
t0 = dbg[0]
print("tstar[0] =", int(eng.tape["tstar"][0]), " prologue(weights->regs) %.2f us, state init %.2f us" % ((dbg[1]-t0)*tick/1e3, (dbg[2]-dbg[1])*tick/1e3))
if dbg[126] > t0:
    print("   reloads of the (value, epoch) pairs, all waves of all sample roles, 6 launches: %d" % dbg[118])
    print("   prep roles of the launch, relative to sample role 0's start: class block 0 done %+.2f us | hw0 block 0 %+.2f | h_x tile 0 %+.2f | last h_x tile %+.2f | sample 0 holds their pairs %+.2f" % tuple((dbg[k] - t0) * tick / 1e3 for k in (123, 127, 124, 125, 126)))
names = ["P1 code->a", "P2 bin->z", "P3 GRU", "P4 A|w_h h", "P5 y+stop+gh_r", "P6 softmax->g+gh_u", "P7 w+gh_n"]
for st in range(int(eng.tape["tstar"][0]) + 1):
    base = 8 + 10 * st
    prev = dbg[base + 9]
    parts = []
    for k in range(9):
        cur = dbg[base + k]
        if cur == 0 or cur < prev: break
        parts.append("%s %.2f" % (names[k], (cur - prev) * tick / 1e3)); prev = cur
    print("step %d: total %.2f us | " % (st, (prev - dbg[base + 9]) * tick / 1e3) + " | ".join(parts))

if dbg[122] > dbg[3]:
    print("   loads + output selection %.2f | log-likelihood sums %.2f | tape flush %.2f | dbar %.2f" % tuple((dbg[x] - dbg[y]) * tick / 1e3 for x, y in ((120, 3), (121, 120), (122, 121), (6, 122))))
if dbg[6] > dbg[3]:
    print("after the loop (log-likelihood sums, dbar, tape flush, output selection): %.2f us" % ((dbg[6] - dbg[3]) * tick / 1e3))
wall_us = (dbg[3] - dbg[0]) * tick / 1e3
print("kernel body %.2f us, shader cycles %d -> shader clock %.0f MHz" % (wall_us, dbg[5] - dbg[4], (dbg[5] - dbg[4]) / wall_us))

bd = dbg[128:]
u = lambda a, b: (bd[b] - bd[a]) * tick / 1e3
print("== k_bwd_conv_fast (sample 0): tape preload %.2f us | staging + output step %.2f | seed bases %.2f | dgpre / dpre bases (MFMA) %.2f | dgpre W_h bases (MFMA) %.2f | wait for the statistics roles + coefficients %.2f | per-step scalars %.2f | tape stores + recurrence %.2f" % (
    u(0, 1), u(1, 8), u(8, 9), u(9, 10), u(10, 11), u(11, 2), u(2, 3), u(3, 4)))
if bd[12]:
    print("   of the wait: until the statistics pairs are fresh %.2f us | coefficients (f64) %.2f us" % (u(11, 12), u(12, 2)))
if bd[48]:
    print("   statistics role 0 (same launch): starts %+.2f us relative to sample role 0 | wait for the baseline roles %.2f us | pairs %.2f us | done %.2f us after sample role 0 started" % (
        (bd[48] - bd[0]) * tick / 1e3, u(48, 51) if bd[51] else 0.0, u(51, 49) if bd[51] else u(48, 49), (bd[50] - bd[0]) * tick / 1e3))
ts0 = int(eng.tape["tstar"][0])
for st in range(ts0, -1, -1):
    base = 16 + 2 * st
    nxt = bd[16 + 2 * (st - 1)] if st > 0 else bd[4]
    print("bwd step %d: cell backward %.2f us | W_hh^T dgh %.2f us" % (st, (bd[base + 1] - bd[base]) * tick / 1e3, (nxt - bd[base + 1]) * tick / 1e3))

for which, nm in ((0, "baseline_rec"), (1, "baseline_sen")):
    d2 = dbg[64 + 32 * which:]
    steps = [(d2[2 + t + 1] - d2[2 + t]) * tick / 1e3 for t in range(9) if d2[2 + t + 1] > d2[2 + t]]
    print("== k_baselines2 %s block(0,0): setup+h_x product %.2f us | steps %s | last step+tail %.2f us | final reduce %.2f us" % (
        nm, (d2[1] - d2[0]) * tick / 1e3, " ".join("%.2f" % v for v in steps), (d2[20] - d2[2 + len(steps)]) * tick / 1e3, (d2[21] - d2[20]) * tick / 1e3))
