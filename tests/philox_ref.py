"""numpy restatement of device_utils.h: philox_uniform -- Philox4x32-10 (Salmon et al., SC'11), counter =
(element index, minibatch counter, stream id, 0), key = 64-bit seed; the first output word's top 24 bits give
a float32 uniform in [0, 1)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox_uniform(seed, c0, c1, c2):
    c0 = np.asarray(c0, np.uint32)
    x0, x1 = c0.copy(), np.full_like(c0, c1, dtype=np.uint32)
    x2, x3 = np.full_like(c0, c2, dtype=np.uint32), np.zeros_like(c0, dtype=np.uint32)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * x0.astype(np.uint64)
            p1 = M1 * x2.astype(np.uint64)
            y0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ x1 ^ k0
            y1 = p1.astype(np.uint32)
            y2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ x3 ^ k1
            y3 = p0.astype(np.uint32)
            x0, x1, x2, x3 = y0, y1, y2, y3
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return (x0 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def conversation_uniforms(seed, minibatch_counter, T, B_global, W, batch_offset=0, batch=None):
    """The draws k_conversation makes for samples [batch_offset, batch_offset + batch) of the global minibatch:
    element (t, b, j) of the z / w streams has index (t*B_global + b)*W + j, the stop stream t*B_global + b."""
    batch = B_global - batch_offset if batch is None else batch
    t = np.arange(T)[:, None, None]; b = (batch_offset + np.arange(batch))[None, :, None]; j = np.arange(W)[None, None, :]
    e = ((t * B_global + b) * W + j).astype(np.uint32)
    u_z = philox_uniform(seed, e, minibatch_counter, 0)
    u_w = philox_uniform(seed, e, minibatch_counter, 2)
    u_s = philox_uniform(seed, (t * B_global + b).astype(np.uint32), minibatch_counter, 1)
    return u_z, u_s, u_w
