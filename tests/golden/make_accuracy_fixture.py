#!/usr/bin/env python
"""Generates tests/golden/accuracy_oracle.json: dev top-6 accuracy of the CPU oracle after N training minibatches on
the learnable synthetic task of tests/test_hip_accuracy.py (SURVEY.md §8d accuracy gate).  The oracle needs minutes
for this on a CPU, so its result is a committed fixture; the GPU test trains the HIP path (a fraction of a second) and
compares.  Both runs are at the plateau of the task by then -- sampled trajectories of two implementations decorrelate
after the first rounding-induced bit flip, so only plateau accuracies are comparable.
usage: python tests/golden/make_accuracy_fixture.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cpu_ref
from tests import test_hip_accuracy as T

fl = cpu_ref.Flags(**T.FLAGS)
desc, draw = T.task()
torch.manual_seed(0)
tape = cpu_ref.UniformTape()
models = cpu_ref.build_agents(fl, rng=tape)
optimizers = cpu_ref.build_optimizers(models, fl)
t0 = time.time()
curve = []
hits = []
for i in range(T.N_TRAIN_MB):
    x, t = draw(T.B)
    u_z, u_s, u_w = cpu_ref.draw_uniforms(fl.max_exchange, T.B, fl.rec_w_dim, seed=100 + i)
    tape.u = {"z": u_z, "s": u_s, "w": u_w}
    tape.t = {"z": 0, "s": 0, "w": 0}
    r = cpu_ref.train_minibatch(models, optimizers, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(desc), fl)
    hits.append(r["hits"])
    if (i + 1) % 250 == 0:
        curve.append(round(100.0 * float(np.mean(hits[-250:])) / T.B, 2))
        print(i + 1, curve[-1], "%.0fs" % (time.time() - t0), flush=True)
for m in models.values():
    m.eval()
xdev, tdev = T.dev_set(draw)
h = 0
for i in range(0, T.N_DEV, T.B):
    h += cpu_ref.eval_batch(models, torch.from_numpy(xdev[i:i + T.B]), torch.from_numpy(tdev[i:i + T.B]), torch.from_numpy(desc), fl)["hits"]
out = dict(n_train_minibatches=T.N_TRAIN_MB, n_dev=T.N_DEV, dev_relabel=T.DEV_RELABEL, oracle_dev_top6_percent=100.0 * h / T.N_DEV,
           oracle_train_top6_percent_per_250=curve, flags=T.FLAGS, torch=torch.__version__, numpy=np.__version__)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "accuracy_oracle.json"), "w"), indent=1)
print(out)
