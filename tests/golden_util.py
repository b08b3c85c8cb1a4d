"""Load tests/golden fixtures (written by oracle/gen_golden.py from the reference)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEPTH_Q = 512.0


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def dequantize_depth(q):
    d = -(q.astype(np.float32) / np.float32(DEPTH_Q))
    d[q == 65535] = -np.inf
    return d


def unpack_bits(bits, g):
    n = bits.shape[0]
    return np.unpackbits(bits, axis=1)[:, : g ** 3].reshape(n, g, g, g).astype(np.float32)


def frames(fx):
    """-> depth_raw [F,N,H,W] f32 (with the special values patched into frame 0), seg f32."""
    d = dequantize_depth(fx["depth_q"])
    if d.ndim == 3:
        d = d[None]
    if len(fx["special_idx"]):
        flat = d[0].reshape(-1)
        flat[fx["special_idx"]] = fx["special_val"]
    seg = fx["seg"].astype(np.float32)
    if seg.ndim == 3:
        seg = seg[None]
    return d, seg
