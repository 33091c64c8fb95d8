"""The parity gate itself (tests/common.py), on the CPU: which gradient mismatches a near-threshold ReLU unit may excuse, and
that the oracle run reports such units."""
import numpy as np

from tests import common

SHAPES = {"receiver": {"y1.weight": (64, 164), "y1.bias": (64,), "y2.weight": (1, 64), "rnn.weight_ih": (192, 32), "w_h.weight": (64, 64)},
          "sender": {"code_layer.weight": (256, 32)},
          "baseline_rec": {"linear1.weight": (500, 96), "linear1.bias": (500,), "linear2.weight": (1, 500), "linear2.bias": (1,)},
          "baseline_sen": {"linear1.weight": (500, 288)}}


def _flips(*per_mb):
    return [dict({"y": set(), "bas_rec": set(), "bas_sen": set(), "where": []}, **f) for f in per_mb]


def _detail(key, flat_rows_cols=None, shape=None):
    """A compare_packed detail whose offending SAMPLE indices decode to the given flat positions of the tensor."""
    if flat_rows_cols is None:
        return (key, np.array([0]), key + " max err 1 (tol 0)")
    stride = max(1, int(np.prod(shape)) // 512)
    return (key, np.array([-(-f // stride) for f in flat_rows_cols]), key + " max err 1 (tol 0)")     # the next SAMPLED position at or after f


def test_entries_fed_by_a_threshold_unit_are_excused_and_nothing_else():
    f = _flips({"y": {7}}, {})
    ok = [_detail("mb0.g.receiver.y1.weight.sample", [7 * 164 + 3], (64, 164)),      # row 7 of y1.weight
          _detail("mb0.g.receiver.y1.bias.sample", [7], (64,)),
          _detail("mb0.g.receiver.y2.weight.sample", [7], (1, 64)),
          _detail("mb0.g.receiver.rnn.weight_ih.sample", [11], (192, 32)),             # through dA -> dh: any entry
          _detail("mb0.g.receiver.y1.weight.norm"), _detail("mb0.gradnorm.receiver"),
          _detail("mb1.p.receiver.w_h.weight.sample", [5], (64, 64))]                  # a LATER minibatch: the parameters moved
    assert common.unexcused_gradient_problems(ok, f, SHAPES) == []
    stride = 64 * 164 // 512
    other_row = 9 * 164                                                                 # (-> the first sampled position of row 9)
    bad = [_detail("mb0.g.receiver.y1.weight.sample", [other_row], (64, 164)),         # row 9 is not on the threshold
           _detail("mb0.g.receiver.w_h.weight.sample", [5], (64, 64)),                 # the message head is not downstream of the y head
           _detail("mb0.g.sender.code_layer.weight.sample", [5], (256, 32)),           # the sender depends on no ReLU
           _detail("mb0.g.baseline_rec.linear1.bias.sample", [3], (500,)),             # no baseline unit flipped
           _detail("mb0.g.receiver.w_h.weight.norm")]
    out = common.unexcused_gradient_problems(bad, f, SHAPES)
    assert len(out) == len(bad), out


def test_baseline_units_and_no_flip_at_all():
    f = _flips({"bas_rec": {3}})
    assert common.unexcused_gradient_problems([_detail("mb0.g.baseline_rec.linear1.bias.sample", [3], (500,)),
                                               _detail("mb0.g.baseline_rec.linear1.weight.sample", [3 * 96], (500, 96))], f, SHAPES) == []
    assert len(common.unexcused_gradient_problems([_detail("mb0.g.baseline_rec.linear2.bias.sample", [0], (1,)),
                                                   _detail("mb0.g.baseline_sen.linear1.weight.norm")], f, SHAPES)) == 2
    assert len(common.unexcused_gradient_problems([_detail("mb0.g.receiver.rnn.weight_ih.norm")], _flips({}), SHAPES)) == 1


def test_oracle_run_reports_threshold_units_per_minibatch():
    """The hooks on the oracle's y1 / baseline linear1 layers: one entry per minibatch, units as column / hidden indices."""
    z, meta = common.load_golden("g3_tiny_adam")
    flips = []
    common.oracle_train_case("g3_tiny_adam", meta, flips=flips)
    assert len(flips) == meta["n_minibatches"]
    for f in flips:
        assert set(f) == {"y", "bas_rec", "bas_sen", "where"} and all(isinstance(r, int) for r in f["y"])
    # moving the threshold up makes units appear: the detector looks at real pre-activations
    old, common.RELU_EPS = common.RELU_EPS, 0.5
    try:
        wide = []
        common.oracle_train_case("g3_tiny_adam", meta, flips=wide)
    finally:
        common.RELU_EPS = old
    assert wide[0]["y"] and wide[0]["bas_rec"] and wide[0]["bas_sen"] and wide[0]["where"]
