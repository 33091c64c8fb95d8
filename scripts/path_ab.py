"""Every specialised kernel path against its fallbacks on the SAME Philox-sampled first minibatch (seed-0 weights, early stopping on):
step counts and top-k hits must be identical (training step AND a following evaluation pass), the six losses and the evaluation's summed log-likelihood within 2e-5 relative (other kernels, other summation order), and a path
run twice must reproduce itself bit for bit.  Round 5 added this after finding k_conv_persist's sender roles running ahead of live
samples (a hand-off that counted stopped samples too generously) -- a bug no oracle test saw because they ran in run-all mode.
usage: path_ab.py [seeds]      exit code 1 on a mismatch"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = {   # name: (engine kwargs over bench.C2, batch, [environment variants])
    "c2 small agents, Adaptive": (dict(), 64, [{}, {"MMG_NO_GAME": "1"}, {"MMG_NO_WGRAD_OPT": "1"}, {"MMG_NO_MERGE_PREP": "1"}, {"MMG_NO_MERGE": "1"}, {"MMG_NO_ROLES": "1"}, {"MMG_NO_FAST": "1"}]),
    "c2 ragged batch 50": (dict(), 50, [{}, {"MMG_NO_GAME": "1"}, {"MMG_NO_FAST": "1"}]),
    "c3 Fixed": (dict(fixed_exchange=True), 64, [{}, {"MMG_NO_MERGE": "1"}, {"MMG_NO_ROLES": "1"}, {"MMG_NO_FAST": "1"}]),
    "c4 W=256 H=1024": (dict(w_dim=256, h_dim=1024), 64, [{}, {"MMG_NO_PERSIST_LL": "1"}, {"MMG_NO_FUSED_S": "1"}, {"MMG_NO_RMSG": "1"}, {"MMG_NO_RSAMPLE": "1"}, {"MMG_NO_PERSIST": "1"}, {"MMG_NO_ROLES": "1"}, {"MMG_NO_TILE": "1"}]),
    "c4 88 samples (two role launches)": (dict(w_dim=256, h_dim=1024), 88, [{}, {"MMG_NO_PERSIST_LL": "1"}, {"MMG_NO_PERSIST": "1"}]),
    "c4 W=128 H=2048 (s1 / s2 roles)": (dict(w_dim=128, h_dim=2048), 48, [{}, {"MMG_NO_RSAMPLE": "1"}, {"MMG_NO_PERSIST": "1"}]),
    "c4 Fixed exchange": (dict(w_dim=256, h_dim=1024, fixed_exchange=True), 64, [{}, {"MMG_NO_PERSIST_LL": "1"}, {"MMG_NO_PERSIST": "1"}]),
    "c4 continuous messages": (dict(w_dim=256, h_dim=1024, use_binary=False, fixed_exchange=True), 64, [{}, {"MMG_NO_PERSIST_LL": "1"}, {"MMG_NO_PERSIST": "1"}]),
    "c4 max_exchange 15": (dict(w_dim=256, h_dim=1024, max_exchange=15), 32, [{}, {"MMG_NO_PERSIST_LL": "1"}, {"MMG_NO_PERSIST": "1"}]),
    "c4 max_exchange 2, 30 samples": (dict(w_dim=256, h_dim=1024, max_exchange=2), 30, [{}, {"MMG_NO_PERSIST_LL": "1"}, {"MMG_NO_PERSIST": "1"}]),
    "c2 max_exchange 15": (dict(max_exchange=15), 64, [{}, {"MMG_NO_GAME": "1"}, {"MMG_NO_FAST": "1"}]),
    "c2 max_exchange 1": (dict(max_exchange=1), 64, [{}, {"MMG_NO_GAME": "1"}, {"MMG_NO_FAST": "1"}]),
    "c4 R=256 wide receiver": (dict(w_dim=256, h_dim=1024, rec_hidden=256), 64, [{}, {"MMG_NO_RC_PERSIST": "1"}, {"MMG_NO_RC_BWD": "1"}, {"MMG_NO_ROLES": "1"}, {"MMG_NO_RC": "1"}]),
    "200 classes, binary, Adaptive": (dict(n_classes=200), 40, [{}, {"MMG_NO_MC": "1"}, {"MMG_TILE": "1"}]),
    "1000 classes, continuous, on the sample tiles (class helpers)": (dict(n_classes=1000, use_binary=False, fixed_exchange=True), 128, [{"MMG_TILE": "1"}, {"MMG_TILE": "1", "MMG_NO_SPLIT": "1"}, {}]),
    "c5 continuous, 512 samples (two tiles per workgroup)": (dict(n_classes=1000, use_binary=False, fixed_exchange=True), 512, [{}, {"MMG_NO_MC3P": "1"}, {"MMG_NO_ROLES": "1"}]),
}
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    import bench
    from multimodalgame_amd.engine import Engine
    from multimodalgame_amd.agents import init_state_dicts
    kw, B = json.loads(sys.argv[2]), int(sys.argv[3])
    cfg = dict(bench.C2); cfg.update(kw)
    out = []
    for seed in range(int(sys.argv[4])):
        eng = Engine(batch=B, **cfg)
        eng.load_state_dicts(init_state_dicts(eng, seed=0))
        feats, target, desc = bench.synthetic_dataset(max(3000, B), cfg["n_classes"], 512, 100)
        dev = eng.device
        x = torch.from_numpy(feats[B * seed:B * seed + B]).to(dev); t = torch.from_numpy(target[B * seed:B * seed + B]).to(dev)
        eng.train_step(x, t, torch.from_numpy(desc).to(dev), seed=11 + seed)
        torch.cuda.synchronize()
        res = [float(v) for v in eng.losses().values()]
        # ... and an evaluation pass (rounded bits, running stop product, no early exit inside the kernels) on the updated weights
        eng.forward(x, t, torch.from_numpy(desc).to(dev), train=False)
        torch.cuda.synchronize()
        eng.check_sync()
        ts = eng.tape["tstar"][:B].cpu().numpy()
        res += [float(ts.sum()), float(eng.tape["hit"][:B].sum().item())]           # (counts: compared exactly)
        res_eval = float(eng.tape["logs"][:B].double().sum().item())
        out.append(res[:6] + [res_eval] + res[6:])
        del eng
    print("RESULT " + json.dumps(out))
    sys.exit(0)
nseeds = sys.argv[1] if len(sys.argv) > 1 else "4"
ok_all = True
for name, (kw, B, variants) in CASES.items():
    res = []
    for env in variants + [variants[0]]:          # (the default path a second time: reproducibility)
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, __file__, "child", json.dumps(kw), str(B), nseeds], env=e, capture_output=True, text=True, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        res.append(json.loads(lines[0][7:]) if lines else None)
        if not lines: print("   child failed:", env, p.stderr.strip().splitlines()[-1:] )
    ref = res[len(variants) - 1]                  # the most generic variant is listed last
    print(name)
    for env, r in zip(variants + [dict(variants[0], again="1")], res):
        tag = " ".join("%s=%s" % kv for kv in env.items()) or "default"
        if r is None or ref is None: print("   %-28s FAILED TO RUN" % tag); ok_all = False; continue
        rel = max(abs(a - b) / max(1.0, abs(b)) for ra, rb in zip(r, ref) for a, b in zip(ra[:6], rb[:6]))
        # (the evaluation pass ROUNDS its message bits: a probability within an ulp of 0.5 may round the other way in another
        #  summation order and the conversation after it differs -- counts must still agree, the summed log-likelihood within 2e-3)
        rel_ev = max(abs(ra[6] - rb[6]) / max(1.0, abs(rb[6])) for ra, rb in zip(r, ref))
        counts = sum(1 for ra, rb in zip(r, ref) if ra[7:] != rb[7:])
        same_as_default = r == res[0]
        good = rel < 2e-5 and rel_ev < 2e-3 and counts == 0 and ("again" not in env or same_as_default)
        ok_all = ok_all and good
        print("   %-28s max relative loss difference %.1e (evaluation pass %.1e), step / hit counts differ in %d of %d seeds%s%s" % (
            tag, rel, rel_ev, counts, len(r), (", reproduces the first run: %s" % same_as_default) if "again" in env else "", "" if good else "   <-- MISMATCH"))
print("OK" if ok_all else "MISMATCH")
sys.exit(0 if ok_all else 1)
