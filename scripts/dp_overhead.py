#!/usr/bin/env python
"""Host/GPU overhead of the data-parallel call sequence on ONE GPU: fused mmg_train_step vs the split
forward/loss_stats/backward/clip_step sequence vs the same with RCCL all-reduces in a single-member group."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.dist import DataParallel
from multimodalgame_amd.agents import init_state_dicts

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
eng = Engine(device=dev, batch=64, **bench.C2)
eng.load_state_dicts(init_state_dicts(eng, seed=0))
feats, target, desc = bench.synthetic_dataset(3000, 30, 512, 100)
x = torch.from_numpy(feats[:64]).to(dev); t = torch.from_numpy(target[:64]).to(dev); d = torch.from_numpy(desc).to(dev)
dp = DataParallel(eng)

def timeit(fn, n=400):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * th / n, 1e6 * (time.perf_counter() - t0) / n

print("fused train_step      host %.1f us  total %.1f us" % timeit(lambda: eng.train_step(x, t, d, seed=1)))
g0 = torch.cuda.CUDAGraph()
s0 = torch.cuda.Stream()
with torch.cuda.stream(s0):
    eng.train_step(x, t, d, seed=1); torch.cuda.synchronize()
    with torch.cuda.graph(g0, stream=s0):
        eng.train_step(x, t, d, seed=1)
print("fused, graph replay   host %.1f us  total %.1f us" % timeit(g0.replay))
dp.world = 1
print("split, no collectives host %.1f us  total %.1f us" % timeit(lambda: dp.train_step(x, t, d, seed=1)))
import numpy as np
acc = {}
for _ in range(20):
    eng.set_profiling(True); dp.train_step(x, t, d, seed=1); torch.cuda.synchronize()
    for name, ms in eng.kernel_times(): acc.setdefault(name, []).append(ms * 1e3)
eng.set_profiling(False)
print("split kernels (us):", {k: round(float(np.mean(v)), 2) for k, v in acc.items()}, "sum %.1f" % sum(float(np.mean(v)) for v in acc.values()))
dp.world = 2
print("split + RCCL (1 rank) host %.1f us  total %.1f us" % timeit(lambda: dp.train_step(x, t, d, seed=1)))
dpd = DataParallel(eng, direct=True); dpd.world = 2
print("split + direct RCCL   host %.1f us  total %.1f us" % timeit(lambda: dpd.train_step(x, t, d, seed=1)))
g = torch.cuda.CUDAGraph()
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): dp.train_step(x, t, d, seed=1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            dp.train_step(x, t, d, seed=1)
    print("graph replay          host %.1f us  total %.1f us" % timeit(g.replay))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
dist.destroy_process_group()
