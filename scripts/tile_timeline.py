"""Per-phase timeline of k_conv_tile for tile 0 (needs libmmg_timing.so: -DMMG_TIMING build).  usage: tile_timeline.py c4|c5|<batch>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalgame_amd import _lib
from multimodalgame_amd import build as _build
_lib.LIB_PATH = _build.build_timing_library()      # -DMMG_TIMING build, compiled on demand (never shipped with the tree)
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
cfg, B, _ = bench.WORKLOADS[which]
cfg = dict(cfg)
eng = Engine(batch=B, **cfg)
eng.load_state_dicts(init_state_dicts(eng, 0))
feats, target, desc = bench.synthetic_dataset(max(3000, B), cfg["n_classes"], 512, 100)
dev = eng.device
x = torch.from_numpy(feats[:B]).to(dev); t = torch.from_numpy(target[:B]).to(dev); d = torch.from_numpy(desc).to(dev)
for it in range(4):
    eng.tape["dbg"].zero_(); eng.tape["dbg2"].zero_()
    eng.train_step(x, t, d, seed=0)
torch.cuda.synchronize()
dbg = eng.tape["dbg"].view(torch.int64).cpu().numpy()
tick = 10.0
print("prologue %.2f us ; loop+tail %.2f us" % ((dbg[1] - dbg[0]) * tick / 1e3, (dbg[2] - dbg[1]) * tick / 1e3))
print("clock64 ticks per us: %.1f" % ((dbg[5] - dbg[4]) / ((dbg[2] - dbg[0]) * tick / 1e3)))
names = ["ph0 s1+gh", "ph1 z", "ph2 gru", "ph3 heads", "ph4 y+softmax", "ph5 dbar", "ph6 g", "ph7 w"]
for st in range(10):
    base = 8 + 16 * st
    if dbg[base] == 0: break
    parts = []
    for k in range(1, 8):
        if dbg[base + k] == 0: break
        if dbg[base + k] < dbg[base + k - 1]: break
        mid = dbg[base + 8 + k - 1]          # after the products of the phase, before its barrier
        parts.append("%s %.2f (prod %.2f)" % (names[k - 1], (dbg[base + k] - dbg[base + k - 1]) * tick / 1e3,
                                             (mid - dbg[base + k - 1]) * tick / 1e3 if mid else -1.0))
    print("step %d: " % st + " | ".join(parts))

if dbg[200]:
    us = lambda a, b: (dbg[b] - dbg[a]) * tick / 1e3
    print("R  signal g(2) at 0; S1: wait-start %.2f -> wait done %.2f | g load %.2f | w gemm+epi+lp %.2f | a gemm+epi %.2f | signal %.2f" % (
        us(222, 200), us(222, 201), us(201, 202), us(202, 203), us(203, 204), us(204, 205)))
    print("S2: wait-start %.2f -> wait done %.2f (S1 signal at %.2f) | a load %.2f | z gemm+epi %.2f | gi gemm+store %.2f | signal %.2f (at %.2f)" % (
        us(222, 210), us(222, 211), us(222, 205), us(211, 212), us(212, 213), us(213, 214), us(214, 215), us(222, 215)))
    print("R: wait-start %.2f -> wait done %.2f" % (us(222, 220), us(222, 221)))

if dbg[250]:
    us = lambda a, b: (dbg[b] - dbg[a]) * tick / 1e3
    print("rs_role (sample 0, step 3), times relative to its signal of step 2: wait-start %.2f | wait done %.2f | gi gather %.2f | GRU .. g %.2f | message %.2f | signal %.2f (at %.2f)" % (
        us(222, 250) if dbg[222] else float("nan"), us(250, 251), us(251, 252), us(252, 253), us(253, 254), us(254, 255), us(250, 255)))
    print("  S1 (tile 0, role 0, step 3): wait-start %.2f -> wait done %.2f | load %.2f | [w gemm] %.2f | a gemm+epi %.2f | signal %.2f ; S2: wait done %.2f | a load %.2f | z gemm+epi %.2f | gi gemm+store %.2f | signal %.2f  (all vs rs_role wait-start)" % (
        us(250, 200), us(250, 201), us(201, 202), us(202, 203), us(203, 204), us(204, 205), us(250, 211), us(211, 212), us(212, 213), us(213, 214), us(214, 215)))

if dbg[170]:
    us = lambda a, b: (dbg[b] - dbg[a]) * tick / 1e3
    base = 8 + 16 * 3
    print("ph0 of step 3 in detail: phase start -> epilogue start %.2f | a = tanh(..) + tape %.2f | pad + zr / c %.2f | gh %.2f" % (us(base, 170), us(170, 171), us(171, 172), us(172, 173)))

if dbg[240]:
    us = lambda a, b: (dbg[b] - dbg[a]) * tick / 1e3
    print("k_send_bwd block (0,0): row list + all loads issued %.2f | statistics -> coefficients %.2f | seeds %.2f | product %.2f | epilogue %.2f" % (us(240, 241), us(241, 242), us(242, 243), us(243, 244), us(244, 245)))

if dbg[224]:
    us = lambda a, b: (dbg[b] - dbg[a]) * tick / 1e3
    print("k_bwd_tile: init %.2f | dy+h* %.2f | dyT+A* %.2f | dA %.2f | dAy + W_hh cache %.2f | loop %.2f" % (
        us(224, 225), us(225, 226), us(226, 227), us(227, 228), us(228, 229), us(229, 230)))
    for t in range(3, -1, -1):
        b = 232 + 4 * t
        if dbg[b] and dbg[b + 1] > dbg[b]:
            nxt = dbg[232 + 4 * (t - 1)] if t > 0 else dbg[230]
            print("  bwd step %d: prefetch + cell backward %.2f | dgh W_hh %.2f" % (t, us(b, b + 1), (nxt - dbg[b + 1]) * tick / 1e3))

# rs_role stamps of every sample that reached step 3 (dbg2[1024 + 8 b + k]: wait-start, pairs / counter seen, gi gathered, g out, message out, signalled)
d2 = eng.tape["dbg2"].view(torch.int64).cpu().numpy()
rs = d2[8192:8192 + 8 * 64].reshape(64, 8)
live = np.nonzero(rs[:, 0])[0]
if len(live):
    u = lambda a: a * tick / 1e3
    r = rs[live]
    print("rs_role at step 3, %d samples: wait %.2f (min %.2f max %.2f) | gather + GRU in %.2f | GRU .. g %.2f | message %.2f | signal %.2f ; last sample in at +%.2f us after the first" % (
        len(live), u((r[:, 1] - r[:, 0]).mean()), u((r[:, 1] - r[:, 0]).min()), u((r[:, 1] - r[:, 0]).max()), u((r[:, 2] - r[:, 1]).mean()), u((r[:, 3] - r[:, 2]).mean()),
        u((r[:, 4] - r[:, 3]).mean()), u((r[:, 5] - r[:, 4]).mean()), u(r[:, 1].max() - r[:, 1].min())))
    if dbg[206] and dbg[201]:
        print("  SA role 0 of tile 0: messages of step 2 seen -> messages of step 3 seen (one step of the ring): %.2f us" % u(dbg[206] - dbg[201]))
    if dbg[215] and dbg[201]:
        print("  tile 0: SB role 0 out at %.2f us before its samples' pairs are seen (mean over tile 0's live samples); SA role 0 sees the messages %.2f us after their mean 'message out'" % (
            u(rs[live[live < 16], 1].mean() - dbg[215]) if (live < 16).any() else float("nan"), u(dbg[201] - rs[live[live < 16], 4].mean()) if (live < 16).any() else float("nan")))
