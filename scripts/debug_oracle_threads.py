"""One-off investigation (not collected by pytest): sensitivity of the CPU oracle to the torch thread count on
g3_continuous (same ReLU-mask flip).  python scripts/debug_oracle_threads.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import common
name = "g3_continuous"
z, meta = common.load_golden(name)
for nt in (128, 128, 32, 8, 1):
    torch.set_num_threads(nt)
    got = common.oracle_train_case(name, meta)
    worst = {}
    for mb in ("mb0", "mb1"):
        e = 0.0
        for k in z.files:
            if k.startswith(mb + ".g.") and k.endswith(".sample"):
                e = max(e, float(np.abs(np.asarray(got[k], np.float64) - np.asarray(z[k], np.float64)).max()))
        worst[mb] = e
    print("threads", nt, worst)
