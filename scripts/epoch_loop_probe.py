"""Where the shipped loop's time per minibatch goes beyond the resident bench: the epoch loop rebuilt piece by piece on one engine
(config 2's shape, synthetic HDF5 of 3000 samples = 46 minibatches per epoch), one device synchronisation per variant."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multimodalgame_amd import misc
from multimodalgame_amd.engine import Engine
from multimodalgame_amd.agents import init_state_dicts
tmp = tempfile.mkdtemp(prefix="mmg_probe_")
paths = misc.write_synthetic_dataset(os.path.join(tmp, "data"), n_classes=30, per_class=100, feat_dim=512, wv_dim=100)
dev = torch.device("cuda", 0)
eng = Engine(device=dev, batch=64, **bench.C2)
eng.load_state_dicts(init_state_dicts(eng, seed=0))
_, _, desc = bench.synthetic_dataset(3000, 30, 512, 100)
d = torch.from_numpy(desc).to(dev)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ep0 = misc.load_epoch(paths["train_file"], 64, 0, True, feats=("avgpool_512",), device=dev)
x0, t0 = ep0.feats["avgpool_512"], ep0.target
def timed(name, body):
    for e in range(5): body(e)
    torch.cuda.synchronize()
    c0 = eng.tape["totals"].cpu().tolist()[0]
    t = time.perf_counter()
    for e in range(E): body(5 + e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    c1 = eng.tape["totals"].cpu().tolist()[0]
    print("%-70s %.2f us / minibatch at %.2f exchange steps" % (name, 1e6 * dt / (E * 46), (c1 - c0) / (E * 46)))
timed("A  one mmg_train_steps(46) per epoch on ONE resident gather", lambda e: eng.train_steps(x0, t0, d, 46, seed=0))
def per_step(e):
    for i in range(46): eng.train_step(x0[64 * i:64 * i + 64], t0[64 * i:64 * i + 64], d, seed=0)
timed("A' 46 mmg_train_step calls per epoch (python loop) on the same gather", per_step)
def with_loader(e):
    ep = misc.load_epoch(paths["train_file"], 64, e, True, feats=("avgpool_512",), device=dev)
    eng.train_steps(ep.feats["avgpool_512"], ep.target, d, 46, seed=0)
timed("B  misc.load_epoch (new permutation + gather) + one mmg_train_steps(46)", with_loader)
def runs_of_23(e):
    ep = misc.load_epoch(paths["train_file"], 64, e, True, feats=("avgpool_512",), device=dev)
    x, t = ep.feats["avgpool_512"], ep.target
    eng.train_steps(x, t, d, 23, seed=0); eng.train_steps(x[23 * 64:], t[23 * 64:], d, 23, seed=0)
timed("C  ... as two runs of 23", runs_of_23)
# ---- the log minibatches of model.run(): a special step (phased, run-all tape) + the log snapshot every 50 steps
from multimodalgame_amd import model as M, flags as F
F.define_flags(); F.FLAGS.Reset(); F.FLAGS(["x", "-model_type", "Adaptive", "-use_binary", "-exchange_samples", "0"]); 
xs, ts = x0[:64], t0[:64]
def special(e, snap):
    eng.train_steps(x0, t0, d, 46, seed=0)
    eng.forward(xs, ts, d, seed=0, train=True, run_all=True); eng.loss_stats(); eng.backward(xs, ts, d); eng.clip_step()
    if snap:
        h = M._log_snapshot_begin(eng, ts, dump=0)
        pend.append(h)
        while len(pend) > 1: M._log_snapshot_end(pend.pop(0))
pend = []
timed("D  46 fused steps + ONE phased run-all step (a log minibatch's update)", lambda e: special(e, False))
timed("E  ... + the log snapshot enqueued (formatted one interval later)", lambda e: special(e, True))
