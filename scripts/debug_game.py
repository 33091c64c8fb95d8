"""One-off: the fused game kernel (kernels_game.h) against the two-launch path on a golden case -- every tape array on the live rows,
the statistics vector, gradients.  python scripts/debug_game.py [philox]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import common
name = "g2_adaptive_c1"
z, meta = common.load_golden(name)
philox = len(sys.argv) > 1 and sys.argv[1] == "philox"
x, target, desc, (u_z, u_s, u_w) = common.case_inputs(meta, 0, name)
res = {}
for tag, env in (("game", None), ("old", "MMG_NO_GAME")):
    if env: os.environ[env] = "1"
    eng = common.make_engine(meta)
    if env: del os.environ[env]
    dev = eng.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if philox:
        eng.train_step(t(x), t(target), t(desc), seed=77)
    else:
        eng.train_step(t(x), t(target), t(desc), t(u_z), t(u_s[..., 0]), t(u_w))
    torch.cuda.synchronize()
    eng.set_profiling(True)
    print(tag, "sync word", int(eng.tape["sync"][511]), "counter", eng.tape["counter"].tolist(), "losses", eng.tape["losses"].tolist())
    res[tag] = ({k: v.detach().cpu().clone() for k, v in eng.tape.items()}, eng.flat_grads.cpu().clone())
A, B = res["game"][0], res["old"][0]
ts = A["tstar"].numpy()
print("tstar equal", np.array_equal(ts, B["tstar"].numpy()), ts[:16], B["tstar"].numpy()[:16])
T, Bn = meta["max_exchange"] if "max_exchange" in meta else 10, ts.shape[0]
live = np.arange(A["z"].shape[0])[:, None] <= ts[None, :]
livew = np.arange(A["z"].shape[0])[:, None] < ts[None, :]
for k in ("z", "pz", "zr", "c", "a", "gru", "s", "ps", "y", "lp_z", "ne_z", "lp_s", "ne_s", "hid_s", "hid_r", "bs", "br", "dls", "dbs", "dbr", "dlz", "dpre", "dgi", "dgh"):
    a, b = A[k].numpy().reshape(live.shape[0], Bn, -1), B[k].numpy().reshape(live.shape[0], Bn, -1)
    d = np.abs(a - b)[live]
    print("%-6s live max diff %.3e" % (k, d.max() if d.size else 0))
for k in ("w", "pw", "g", "lp_w", "ne_w", "dbar", "dlw", "dgpre"):
    a, b = A[k].numpy().reshape(live.shape[0], Bn, -1), B[k].numpy().reshape(live.shape[0], Bn, -1)
    d = np.abs(a - b)[livew]
    print("%-6s live(w) max diff %.3e" % (k, d.max() if d.size else 0))
hl = np.arange(A["h"].shape[0])[:, None] <= (ts[None, :] + 1)
d = np.abs(A["h"].numpy() - B["h"].numpy())[hl]; print("h max diff %.3e" % d.max())
for k in ("outp", "dist", "sm", "logs", "hit", "dy", "Astar", "dA", "hstar", "dhx", "dC", "Py2", "stats", "rmap", "rcount", "basehx", "Cd", "hx"):
    a, b = A[k].numpy().astype(np.float64), B[k].numpy().astype(np.float64)
    print("%-6s max diff %.3e" % (k, np.abs(a - b).max()))
sa, sb = A["stats"].numpy(), B["stats"].numpy()
bad = np.nonzero(np.abs(sa - sb) > 1e-6)[0]
print("stats differing entries", bad[:40], sa[bad[:10]], sb[bad[:10]])
g = (res["game"][1] - res["old"][1]).abs()
print("grads max diff %.3e at %d of %d" % (g.max(), int(g.argmax()), g.numel()))
dz = np.abs(A["z"].numpy() - B["z"].numpy()) * live[:, :, None]
idx = np.argwhere(dz > 0)
print("z diffs:", len(idx), "first", idx[:12].tolist())
print("by t:", np.bincount(idx[:, 0], minlength=10).tolist(), "by j:", np.bincount(idx[:, 2], minlength=32).tolist())
t0, b0 = idx[0][0], idx[0][1]
print("game z", A["z"][t0, b0].tolist()); print("old  z", B["z"][t0, b0].tolist()); print("game zr next", A["zr"][min(t0 + 1, 9), b0].tolist())
print("game w ", A["w"][t0, b0].tolist())
