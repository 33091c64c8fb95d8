"""Seed-to-seed spread of the dev top-6 accuracy of the GPU path on the noisy synthetic task (choosing the accuracy gate's noise level)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu_ref
from tests import common
from tests import test_hip_accuracy as T

def run(sigma_dev, seed, n_mb, n_dev, lr, sigma=0.3):
    fl = cpu_ref.Flags(**dict(T.FLAGS, learning_rate=lr))
    meta = dict(fl.__dict__); meta.update(n_classes=T.D, batch=T.B, n_minibatches=n_mb, seed_weights=3, seed_data=0, seed_uniforms=100)
    rs = np.random.RandomState(77)
    proto = rs.standard_normal((T.D, T.F)).astype(np.float32)
    desc = (0.3 * rs.standard_normal((T.D, T.V))).astype(np.float32)
    def draw(n):
        t = rs.randint(0, T.D, size=(n,)).astype(np.int64)
        return np.abs(proto[t] + sigma * rs.standard_normal((n, T.F))).astype(np.float32), t
    torch.manual_seed(0)
    models = cpu_ref.build_agents(fl, rng=cpu_ref.UniformTape())
    eng = common.make_engine(meta)
    eng.load_state_dicts({a: {k: v.detach().clone() for k, v in m.state_dict().items()} for a, m in models.items()})
    dev = eng.device; dd = torch.from_numpy(desc).to(dev)
    for i in range(n_mb):
        x, t = draw(T.B)
        eng.train_step(torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev), dd, seed=seed)
    def draw_dev(n):
        t = rs.randint(0, T.D, size=(n,)).astype(np.int64)
        return np.abs(proto[t] + sigma_dev * rs.standard_normal((n, T.F))).astype(np.float32), t
    xdev, tdev = draw_dev(n_dev); hits = 0
    for i in range(0, n_dev, T.B):
        eng.forward(torch.from_numpy(xdev[i:i + T.B]).to(dev), torch.from_numpy(tdev[i:i + T.B]).to(dev), dd, train=False, run_all=True)
        hits += int(eng.tape["hit"].sum().item())
    return 100.0 * hits / n_dev

for sigma in (float(a) for a in sys.argv[1].split(",")):
    for n_mb in (3000,):
        accs = [run(sigma, s, n_mb, 30080, 1e-3) for s in (2024, 7, 99)]
        print("sigma %.2f n_mb %d: %s  spread %.2f" % (sigma, n_mb, ["%.2f" % a for a in accs], max(accs) - min(accs)), flush=True)
