import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions: the max abs error per compared quantity (tests/common.py: MAXERR) -> tests/out/parity_maxerr.json."""
    try:
        from tests import common
    except Exception:
        return
    if not common.MAXERR:
        return
    import json
    out = os.path.join(REPO, "tests", "out")
    os.makedirs(out, exist_ok=True)
    worst_fwd = max([v for k, v in common.MAXERR.items() if not common.is_grad_key(k)] or [0.0])
    with open(os.path.join(out, "parity_maxerr.json"), "w") as f:
        json.dump({"note": "max |HIP - reference| per (case:quantity) over this pytest session; forward quantities are gated at "
                           "atol 1e-4, rtol 0 (tests/common.py: compare_packed)",
                   "worst_forward": worst_fwd, "entries": dict(sorted(common.MAXERR.items()))}, f, indent=1)
