"""The parity gate itself (tests/common.py), on the CPU: a gradient mismatch is never excused -- the oracle is re-run with the
near-threshold ReLU units forced to the side the other implementation put them on, and must then agree everywhere."""
import numpy as np
import pytest

from tests import common

NAME = "g3_tiny_adam"


class _Eng(object):
    """Stand-in for the GPU engine: only what assert_parity reads (the other implementation's pre-activations)."""
    relu_capture = None


def _oracle_capture(meta, name, flips):
    """A capture that agrees with the oracle on every near-threshold unit (both implementations on the same side)."""
    fl = common.flags_from_meta(meta)
    B, D, R, K, T = meta["batch"], meta["n_classes"], fl.rec_hidden, fl.baseline_hid_dim, fl.max_exchange
    cap = []
    for f in flips:
        c = dict(y_pre={}, hid_r=np.zeros((T, B, K)), hid_s=np.zeros((T, B, K)))
        for p in f["pos"]:
            if p[0] == "y":
                c["y_pre"][(p[2], p[3], p[4])] = 1.0 if p[-1] else -1.0
            else:
                c["hid_r" if p[0] == "bas_rec" else "hid_s"][p[1], p[2], p[3]] = 1.0 if p[-1] else 0.0
        cap.append(c)
    return cap


@pytest.fixture()
def wide_eps():
    old, common.RELU_EPS = common.RELU_EPS, 0.02          # so that the tiny case HAS near-threshold units
    yield
    common.RELU_EPS = old


def test_oracle_run_reports_threshold_units_per_minibatch(wide_eps):
    """The hooks on the oracle's y1 / baseline linear1 layers: one entry per minibatch, units with position and side."""
    z, meta = common.load_golden(NAME)
    flips = []
    common.oracle_train_case(NAME, meta, flips=flips)
    assert len(flips) == meta["n_minibatches"]
    for f in flips:
        assert {"y", "bas_rec", "bas_sen", "where", "pos", "case"} <= set(f)
    assert flips[0]["pos"] and any(p[0] == "y" for f in flips for p in f["pos"])
    common.RELU_EPS = 2e-5
    tight = []
    common.oracle_train_case(NAME, meta, flips=tight)
    assert sum(len(f["pos"]) for f in tight) < sum(len(f["pos"]) for f in flips)      # the detector looks at real pre-activations


def test_forced_masks_only_lists_units_on_the_other_side(wide_eps):
    z, meta = common.load_golden(NAME)
    flips = []
    common.oracle_train_case(NAME, meta, flips=flips)
    cap = _oracle_capture(meta, NAME, flips)
    assert not any(common.forced_masks(flips, cap))                                    # same side everywhere: nothing to force
    p = next(p for p in flips[0]["pos"] if p[0] == "y")
    cap[0]["y_pre"][(p[2], p[3], p[4])] *= -1.0                                        # the other implementation flipped ONE unit
    forced = common.forced_masks(flips, cap)
    assert forced[0] == {p[:-1] + (not p[-1],)} and not any(forced[1:])


def test_a_flipped_unit_is_explained_only_by_the_forced_rerun(wide_eps):
    common.RELU_EPS = 5e-4        # (the closest unit of this tiny case sits at |pre| = 3.7e-4: forcing it moves the logits by < 1e-4)
    _flipped_unit_case()


def _flipped_unit_case():
    """`got` = the oracle with one near-threshold y-head unit on the other side (what a GPU run with another summation order
    looks like).  With the capture saying so, assert_parity re-runs the oracle with that unit forced and passes; with a
    capture that claims the GPU agrees with the oracle on every unit the same mismatch FAILS (no blanket excuse), and so
    does a mismatch in a tensor the unit does not feed."""
    z, meta = common.load_golden(NAME)
    flips = []
    want = common.oracle_train_case(NAME, meta, flips=flips)
    cap = _oracle_capture(meta, NAME, flips)
    # the unit with the largest effect: try the y-head units of minibatch 0 until the gradients really move
    for p in [p for p in flips[0]["pos"] if p[0] == "y"]:
        force = [set() for _ in flips]
        force[0] = {p[:-1] + (not p[-1],)}
        got = common.oracle_train_case(NAME, meta, force=force)
        if common.compare_packed(got, want, atol=1e-4, rtol=1e-3, skip=("y2.bias",), shift_invariant=True):
            break
    else:
        pytest.skip("no near-threshold unit of this case moves a gradient beyond the tolerance")
    eng = _Eng()
    cap[0]["y_pre"][(p[2], p[3], p[4])] *= -1.0
    eng.relu_capture = cap
    common.assert_parity(got, want, flips, eng, None, skip=("y2.bias",))              # explained by the forced re-run
    eng.relu_capture = _oracle_capture(meta, NAME, flips)                             # "the GPU is on the oracle's side"
    with pytest.raises(AssertionError):
        common.assert_parity(got, want, flips, eng, None, skip=("y2.bias",))
    bad = dict(got)
    k = next(k for k in bad if ".g.sender.code_layer.weight" in k)
    bad[k] = np.asarray(bad[k]) + 1.0                                                 # an error no ReLU unit can explain
    eng.relu_capture = cap
    with pytest.raises(AssertionError):
        common.assert_parity(bad, want, flips, eng, None, skip=("y2.bias",))


def test_forward_gate_is_absolute_and_losses_beyond_fp32_resolution_are_gated_against_float64():
    want = {"mb0.losses": np.array([600.0, 0.5]), "mb0.logs": np.array([80.0])}
    got = {"mb0.losses": np.array([600.0005, 0.5]), "mb0.logs": np.array([80.0])}
    assert common.compare_packed(got, want, label="config2")                          # 5e-4 on a loss: fails, whatever the label ...
    assert common.compare_packed(got, want, label="config4-b88")
    # ... unless the float64 oracle says the fp32 oracle itself is that far from the exact value: |got - f64| <= |want - f64| + 1e-4
    f64 = {"mb0.losses": np.array([600.00045, 0.5])}
    assert not common.compare_packed(got, want, label="config4-b88", f64=f64)         # got is 5e-5 from exact, the fp32 oracle 4.5e-4
    assert common.compare_packed(got, want, label="config4-b88", f64={"mb0.losses": np.array([599.9999, 0.5])})   # got 6e-4 from exact, the oracle 1e-4
    got["mb0.logs"] = np.array([80.0005])
    assert common.compare_packed(got, want, label="config4-b88", f64=f64)             # only the entries f64 holds are gated that way
    got = {"mb0.losses": np.array([600.0, 0.5002]), "mb0.logs": np.array([80.0])}
    assert common.compare_packed(got, want, label="config4", f64=f64)                 # a small entry: 2e-4 from exact fails
