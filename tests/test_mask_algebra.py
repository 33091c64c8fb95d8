"""Property test of the reformulation the kernels rely on (kernels_bwd.h: k_stats): the reference's mask lists
(model.py:775, 852, 866, 870, 1248-1262) are fully described by t*(b), the step whose logits are the output:
    m_t[b] == 1 (binary_s / binary_sen / bas masks)        <=>  t <= t*(b)
    m_{t+1}[b] == 1 after the forced final zero (binary_rec) <=>  t <  t*(b)
    y_mask_t[b] == 1                                         <=>  t == t*(b)   (exactly one per sample)
and the number of executed steps is max_b t*(b) + 1 when every sample stops, else max_exchange."""
import numpy as np
from hypothesis import given, settings, strategies as st


def reference_masks(s_bits, max_exchange):
    """Literal mask bookkeeping of exchange() with break_early=True.  s_bits: [T, B] in {0,1}."""
    B = s_bits.shape[1]
    masks = [np.ones(B, np.uint8)]
    n = 0
    for t in range(max_exchange):
        masks.append(np.minimum(masks[-1], s_bits[t].astype(np.uint8)))      # model.py:852
        n += 1
        if masks[-1].sum() == 0:                                             # model.py:866
            break
    masks[-1][:] = 0                                                         # model.py:870
    return masks, n


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 8), st.integers(1, 12), st.integers(0, 2**32 - 1), st.floats(0.05, 0.95))
def test_tstar_describes_every_mask(T, B, seed, p_continue):
    rs = np.random.RandomState(seed)
    s_bits = (rs.rand(T, B) < p_continue).astype(np.uint8)
    masks, n = reference_masks(s_bits, T)
    # t*(b): first step with s == 0, else the last step  (k_conversation: s_misc[1])
    tstar = np.array([next((t for t in range(T) if s_bits[t, b] == 0), T - 1) for b in range(B)])
    assert n == (tstar.max() + 1 if (s_bits.min(0) == 0).all() else T)
    binary_s, binary_rec = masks[:-1], masks[1:-1]                           # model.py:1256-1257
    y_masks = [np.minimum(1 - m1, m2) for m1, m2 in zip(masks[1:], masks[:-1])]   # model.py:1261
    for t in range(n):
        np.testing.assert_array_equal(binary_s[t], (t <= tstar).astype(np.uint8))
        np.testing.assert_array_equal(y_masks[t], (t == tstar).astype(np.uint8))
        if t < n - 1:
            np.testing.assert_array_equal(binary_rec[t], (t < tstar).astype(np.uint8))
    assert (np.sum(y_masks, 0) == 1).all()                                   # the -debug assertion at model.py:898-900
    # steps the reference never ran contribute nothing: no sample is active there
    assert (tstar < n).all()
