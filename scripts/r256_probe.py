import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, time
import bench
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
cfg = dict(bench.WORKLOADS["c4r256"][0])
eng = Engine(batch=64, **cfg)
eng.load_state_dicts(init_state_dicts(eng, seed=0))
feats, target, desc = bench.synthetic_dataset(3000, 30, 512, 100)
dev = eng.device
x = torch.from_numpy(feats[:64]).to(dev); t = torch.from_numpy(target[:64]).to(dev); d = torch.from_numpy(desc).to(dev)
for _ in range(5): eng.train_step(x, t, d, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): eng.train_step(x, t, d, seed=1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
eng.set_profiling(True); eng.train_step(x, t, d, seed=1); torch.cuda.synchronize()
print("c4 R=256: %.1f us/minibatch" % (dt * 1e6), {k: round(v * 1e3, 1) for k, v in eng.kernel_times()})
eng.check_sync()
