#!/usr/bin/env python
"""Runs N data-parallel-style (split) or fused training steps for a kernel trace.  usage: split_trace.py split|fused|direct [n]"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.dist import DataParallel
from multimodalgame_amd.agents import init_state_dicts
mode = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
if mode in ("direct", "c10d"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1)
eng = Engine(device=dev, batch=64, **bench.C2)
eng.load_state_dicts(init_state_dicts(eng, seed=0))
feats, target, desc = bench.synthetic_dataset(3000, 30, 512, 100)
x = torch.from_numpy(feats[:64]).to(dev); t = torch.from_numpy(target[:64]).to(dev); d = torch.from_numpy(desc).to(dev)
dp = DataParallel(eng, direct=(mode == "direct"))
dp.world = 2 if mode in ("direct", "c10d") else 1
for i in range(n):
    if mode == "fused": eng.train_step(x, t, d, seed=1)
    else: dp.train_step(x, t, d, seed=1)
torch.cuda.synchronize()
