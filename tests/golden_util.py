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


# --------------------------------------------------------------------------- #
# deterministic, platform-independent network weights for the policy fixtures:
# pure integer / float64 arithmetic, so the generator (reference side) and the
# tests (our side) build bit-identical parameters without storing 4.6 MB.
# --------------------------------------------------------------------------- #
def det_array(shape, seed, scale):
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    h = (idx * np.uint64(2654435761) + np.uint64(seed) * np.uint64(40503) + np.uint64(12345)) % np.uint64(2 ** 32)
    h = (h ^ (h >> np.uint64(15))) * np.uint64(2246822519) % np.uint64(2 ** 32)
    h = (h ^ (h >> np.uint64(13))) * np.uint64(3266489917) % np.uint64(2 ** 32)
    h = h ^ (h >> np.uint64(16))
    u = h.astype(np.float64) / 4294967296.0 - 0.5
    return (u * 2.0 * scale).astype(np.float32).reshape(shape)


def det_state_dict(shapes):
    """shapes: ordered dict name -> shape (from model.state_dict()). Weights ~ U(-s, s) with
    s = sqrt(3 / fan_in) (unit-variance-preserving), BN weight ~ 1 +- 0.2, running_var in [0.5, 1.5]."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, np.int64)
        elif name.endswith("running_var"):
            out[name] = (1.0 + det_array(shape, 1000 + i, 0.5)).astype(np.float32)
        elif name.endswith("running_mean"):
            out[name] = det_array(shape, 1000 + i, 0.3)
        elif len(shape) == 1 and ("1.weight" in name or "4.weight" in name):  # BatchNorm affine weight
            out[name] = (1.0 + det_array(shape, 1000 + i, 0.2)).astype(np.float32)
        elif len(shape) == 1:
            out[name] = det_array(shape, 1000 + i, 0.1)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = det_array(shape, 1000 + i, float(np.sqrt(3.0 / fan_in)))
    return out
