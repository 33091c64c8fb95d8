"""us per training minibatch of config 4's agents with rec_hidden 256 (wide receiver, kernels_rc.h) over the batch size: up to 4 tiles the
conversation is ONE launch of co-resident roles (k_rc_persist), beyond that the per-step launches (k_rc_gru / k_rc_heads / k_rc_query)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
cfg = dict(bench.WORKLOADS["c4r256"][0])
feats, target, desc = bench.synthetic_dataset(3000, 30, 512, 100)
for B in (16, 32, 64, 80, 128, 256, 512):
    eng = Engine(batch=B, **cfg)
    eng.load_state_dicts(init_state_dicts(eng, seed=0))
    dev = eng.device
    x = torch.from_numpy(feats[:B]).to(dev); t = torch.from_numpy(target[:B]).to(dev); d = torch.from_numpy(desc).to(dev)
    for _ in range(5): eng.train_step(x, t, d, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): eng.train_step(x, t, d, seed=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    eng.set_profiling(True); eng.train_step(x, t, d, seed=1); torch.cuda.synchronize()
    kt = {k: round(v * 1e3, 1) for k, v in eng.kernel_times()}
    eng.check_sync()
    print("B=%4d  %8.1f us/minibatch  %8.0f samples/s  %s" % (B, dt * 1e6, B / dt, kt), flush=True)
    del eng
