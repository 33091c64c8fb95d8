#!/usr/bin/env python
"""us per training minibatch of the other BASELINE configs on one GPU (they are parity-test cases, not bench lines)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts

CASES = {
    "c2 adaptive B=64": dict(bench.C2),
    "c3 fixed, one 64-sample shard": dict(bench.C2, fixed_exchange=True),
    "c3 fixed B=512 on one GPU": dict(bench.C2, fixed_exchange=True, batch=512),
    "c4 W=256 H=1024 B=64": dict(bench.C2, w_dim=256, h_dim=1024),
    "c4 with R=256 (wide receiver)": dict(bench.C2, w_dim=256, h_dim=1024, rec_hidden=256),
    "c5 continuous D=1000 B=2048": dict(bench.C2, use_binary=False, fixed_exchange=True, n_classes=1000, batch=2048),
}
only = sys.argv[1] if len(sys.argv) > 1 else None      # e.g. 'c4': run just the cases whose name starts with it
for name, cfg in CASES.items():
    if only and not name.startswith(only):
        continue
    B = cfg.pop("batch", 64)
    eng = Engine(batch=B, **cfg)
    eng.load_state_dicts(init_state_dicts(eng, seed=0))
    feats, target, desc = bench.synthetic_dataset(max(3000, B), cfg["n_classes"], 512, 100)
    dev = eng.device
    x = torch.from_numpy(feats[:B]).to(dev); t = torch.from_numpy(target[:B]).to(dev); d = torch.from_numpy(desc).to(dev)
    n = int(os.environ.get("N", "50"))
    for _ in range(5): eng.train_step(x, t, d, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.train_step(x, t, d, seed=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    eng.set_profiling(True); eng.train_step(x, t, d, seed=1); torch.cuda.synchronize()
    kt = {k: round(v * 1e3, 1) for k, v in eng.kernel_times()}
    eng.set_profiling(False)
    print("%-34s %9.1f us/minibatch  %8.0f samples/s   %s" % (name, dt * 1e6, B / dt, kt), flush=True)
    del eng
