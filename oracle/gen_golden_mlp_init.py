"""Generate the seeded-initialisation golden of the MLP extractor by RUNNING THE REFERENCE (this container only).

TEST INFRASTRUCTURE ONLY.   python oracle/gen_golden_mlp_init.py

  F16_mlp_init   the reference's own MlpExtractor (stable_baselines3/common/torch_layers.py) constructed behind torch.manual_seed(3) for
                 three architectures: the parameters' default initialisation draws from torch's global generator in the order the layers
                 are CREATED (policy layer i, value layer i, depth by depth), so the values pin that order -- what ortho_init=False
                 policies depend on.  Stored: every state-dict tensor.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ARCHS = {"default": [dict(pi=[64, 64], vf=[64, 64])], "shared": [96, dict(pi=[48], vf=[32, 16])], "ragged": [32, dict(pi=[8, 9, 10], vf=[7])]}
FEATURES, SEED = 40, 3


def main():
    ref_harness.import_reference()
    import importlib
    tl = importlib.import_module("stable_baselines3.common.torch_layers")
    out = {"feature_dim": np.int64(FEATURES), "seed": np.int64(SEED)}
    for tag, arch in ARCHS.items():
        torch.manual_seed(SEED)
        m = tl.MlpExtractor(FEATURES, arch, torch.nn.Tanh, "cpu")
        for k, v in m.state_dict().items():
            out[f"{tag}/{k}"] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", "F16_mlp_init.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
