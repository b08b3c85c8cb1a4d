"""Minimal observation / action space descriptions (gym is not a dependency).

Only what the reference algorithm reads from the env: `observation_space.shape`,
`action_space.nvec` / `.shape` (env_train_gennbv.py:459-492, wrapper :59-88)."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low, high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)
        if shape is None:
            shape = low.shape if low.ndim else high.shape
        self.shape = tuple(int(s) for s in shape)
        with np.errstate(all="ignore"):
            self.low = np.broadcast_to(low, self.shape).astype(dtype)
            self.high = np.broadcast_to(high, self.shape).astype(dtype)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return f"Box{self.shape}"


class MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)

    def __repr__(self):
        return f"MultiDiscrete({self.nvec.tolist()})"
