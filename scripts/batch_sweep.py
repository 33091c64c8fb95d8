#!/usr/bin/env python
"""Throughput of one training minibatch against the batch size, per agent shape (one GPU): which kernels the library
picks at each size and how far the MFMA-shaped ones get when the problem is large enough.
usage: batch_sweep.py [c2|c4|c5] [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
batches = [int(a) for a in sys.argv[2:]] or [64, 256, 1024, 4096]
cfg0, _, _ = bench.WORKLOADS[which]
for B in batches:
    cfg = dict(cfg0)
    eng = Engine(batch=B, **cfg)
    eng.load_state_dicts(init_state_dicts(eng, seed=0))
    feats, target, desc = bench.synthetic_dataset(max(3000, B), cfg["n_classes"], 512, 100)
    dev = eng.device
    x = torch.from_numpy(feats[:B]).to(dev); t = torch.from_numpy(target[:B]).to(dev); d = torch.from_numpy(desc).to(dev)
    n = 30
    for _ in range(5): eng.train_step(x, t, d, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.train_step(x, t, d, seed=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    eng.check_sync()
    eng.set_profiling(True); eng.train_step(x, t, d, seed=1); torch.cuda.synchronize()
    kt = {}
    for k, v in eng.kernel_times(max_kernels=512): kt[k] = kt.get(k, 0.0) + v * 1e3
    eng.set_profiling(False)
    tstar = eng.tape["tstar"].float().mean().item() + 1.0
    flops = 0.0
    for k in kt:
        bound, amt = bench.algorithmic_work(k, cfg, B, tstar)
        if bound == "mfma": flops += amt
    print("%s B=%-5d %9.1f us/minibatch  %9.0f samples/s  %6.2f TFLOP/s over the MFMA-shaped kernels (%.0f MFLOP)   %s" % (
        which, B, dt * 1e6, B / dt, flops / dt / 1e12, flops / 1e6, {k: round(v, 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1])}), flush=True)
    del eng
