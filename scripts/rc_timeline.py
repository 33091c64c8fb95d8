"""In-kernel timeline of k_rc_persist (wide receiver, config 4 with R = 256): tile 0, exchange step 3 (needs the -DMMG_TIMING build,
compiled on demand).  Stamps: kernels_rc.h MMG_RSTAMP slots 100.. (S1 role 0), 110.. (S2 role 0), 120.. (RC role 0), 140.. (RC role 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalgame_amd import _lib
from multimodalgame_amd import build as _build
_lib.LIB_PATH = _build.build_timing_library()
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
import bench
cfg, B, _ = bench.WORKLOADS["c4r256"]
eng = Engine(batch=B, **dict(cfg))
eng.load_state_dicts(init_state_dicts(eng, 0))
feats, target, desc = bench.synthetic_dataset(3000, cfg["n_classes"], 512, 100)
dev = eng.device
x = torch.from_numpy(feats[:B]).to(dev); t = torch.from_numpy(target[:B]).to(dev); d = torch.from_numpy(desc).to(dev)
for it in range(4):
    eng.train_step(x, t, d, seed=0)
torch.cuda.synchronize()
dbg = eng.tape["dbg"].view(torch.int64).cpu().numpy()
t0 = min(int(v) for v in dbg[100:160] if v)
def delta(name, a, slots, labels):
    prev = int(dbg[a]); out = []
    for sl, l in zip(slots, labels):
        if dbg[sl]:
            out.append("%s +%.2f" % (l, (int(dbg[sl]) - prev) * 10.0 / 1e3)); prev = int(dbg[sl])
    print(name + ": " + " | ".join(out))
us = lambda s: (int(dbg[s]) - t0) * 10.0 / 1e3 if dbg[s] else float("nan")
def row(name, base, labels):
    print(name + ": " + " | ".join("%s %.2f" % (l, us(base + k)) for k, l in enumerate(labels)))
row("S1 role 0 ", 100, ["wait w", "got", "mfma", "stored", "signalled a"])
row("S2 role 0 ", 110, ["wait a", "got", "mfma", "stored", "signalled z"])
rc = ["wait z", "got", "gru", "sig h", "extras", "got h", "heads", "sig y", "got y", "query", "sig w"]
row("RC role 0 ", 120, rc)
row("RC role 5 ", 140, rc)
delta("role 0 gru   (from 'got z')", 121, [160, 161, 122], ["loads+mfma", "acc sync", "cell+stores"])
delta("role 0 heads (from 'got h')", 125, [165, 166, 167, 168, 126], ["loads+mfma", "A / w_h h", "partial logits", "stop bit", "alive"])
delta("role 0 query (from 'got y')", 128, [170, 171, 172, 173, 174, 175, 129], ["operands", "logits", "softmax", "mixture", "h_w", "message mfma", "sample+stores"])
