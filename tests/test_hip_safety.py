"""A timed-out in-launch dependency wait (device_utils.h: role_wait -> sync[511]) must never train on silently:
k_opt skips the update of that minibatch and the next mmg_train_step fails."""
import numpy as np
import pytest
import torch

from multimodalgame_amd import _lib
from tests import common

pytestmark = pytest.mark.gpu


def test_dependency_timeout_skips_update_and_raises():
    z, meta = common.load_golden("g2_adaptive_c1")
    eng = common.make_engine(meta)
    dev = eng.device
    x, target, desc, _ = common.case_inputs(meta, 0)
    xd, td, dd = [torch.from_numpy(a).to(dev) for a in (x, target, desc)]
    eng.train_step(xd, td, dd, seed=3)                       # a healthy step changes the parameters
    torch.cuda.synchronize()
    before = eng.flat_params.clone()
    state_before = eng.opt_state.clone()
    eng.tape["sync"][511] = 2                                # what role_wait stores when dependency 1 times out
    eng.train_step(xd, td, dd, seed=3)                       # the error word is set: k_opt must leave everything untouched
    torch.cuda.synchronize()
    torch.testing.assert_close(eng.flat_params, before, rtol=0, atol=0)
    torch.testing.assert_close(eng.opt_state, state_before, rtol=0, atol=0)
    with pytest.raises(_lib.MmgError, match="timed out"):    # ... and the following call reports it (no host sync involved)
        eng.train_step(xd, td, dd, seed=3)
    with pytest.raises(_lib.MmgError):
        eng.check_sync()
