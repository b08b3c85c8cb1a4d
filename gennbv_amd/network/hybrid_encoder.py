"""Hybrid_Encoder: the multi-source (pose history + occupancy grid) features extractor.

Drop-in for `gennbv/network/hybrid_encoder.py:18-98` behind the SB3 features-extractor
protocol (constructor kwargs of gennbv/train/train_gennbv.py:152-168 unchanged, `.features_dim`,
`forward(observations[B, D_obs]) -> [B, features_dim]`), with the reference's
`state_dict` key names so its checkpoints load unchanged:

    naive_encoder_grid.{0,1,3,4}.*   Conv3d(1,16,k3,s2) BN Conv3d(16,16,k3,s2) BN
    output_layer_grid.0.*            Linear(16*o2^3, 256)
    naive_encoder_action.{0,2}.*     Linear(stack*24, 256), Linear(256, 256)
    output_layer.0.*                 Linear(512, 256)

The modules only HOLD the parameters (torch layouts, reference keys); every forward / backward
runs on the hand-written gfx950 kernels through the C-ABI (gennbv_amd/ops/encoder_ops.py).  There is
no second implementation in the product: a CPU tensor is refused (`GennbvHipError`).  The plain-torch
restatement of the reference forward that the floating-point parity tests compare against lives in
tests/torch_reference.py.

Additive: the grid edge G is inferred from `observation_space` (flat row = state | G^3 | two 64x64 gray
frames, env_wrapper_gennbv_train.py:104-110) or given as `grid_size`; the reference hard-codes
20^3 / 8000 / 1024 (hybrid_encoder.py:47,90-91) and G=20 reproduces it exactly.
The released encoder never reads the `state_rgb` slice of the observation (:76-98); neither does this one.
"""
from __future__ import annotations

import copy
from typing import List, Optional, Tuple

import torch
from torch import nn

RGB_DIM = 2 * 64 * 64  # obs["state_rgb"]: the two most recent 64x64 gray frames (env_train_gennbv.py:363)


def conv_out(g: int) -> Tuple[int, int]:
    o1 = (g - 3) // 2 + 1
    return o1, (o1 - 3) // 2 + 1


def infer_grid_size(observation_space, state_dim: int) -> int:
    """G from the flat observation width; 20 (the reference's literal) when the space does not say."""
    shape = getattr(observation_space, "shape", None)
    if not shape:
        return 20
    cells = int(shape[-1]) - int(state_dim) - RGB_DIM
    g = round(max(cells, 0) ** (1.0 / 3.0))
    for c in (g - 1, g, g + 1):
        if c > 0 and c ** 3 == cells:
            return c
    return 20


class Hybrid_Encoder(nn.Module):
    backend = "hip"  # the only implementation of the product class (tests/torch_reference.py subclasses it as "torch")

    def __init__(self, observation_space, encoder_param=None, net_param=None, visual_input_shape=None,
                 state_input_shape=None, grid_size: Optional[int] = None, compute_dtype=torch.float32):
        assert encoder_param is not None, "Need parameters !"
        assert net_param is not None, "Need parameters !"
        assert isinstance(visual_input_shape, (List, Tuple, list, tuple)), "Use tuple or list"
        assert isinstance(state_input_shape, (List, Tuple, list, tuple)), "Use tuple or list"
        super().__init__()
        self.image_channel = visual_input_shape[0]
        self.image_shape = visual_input_shape[1:]
        self.state_input_shape = state_input_shape
        # the reference pops the last entry of the CALLER's list in place (hybrid_encoder.py:32-33), which breaks a
        # second construction from the same policy_kwargs (save -> load -> save); read it without mutating
        self._features_dim = copy.deepcopy(net_param)["append_hidden_shapes"][-1]
        self._observation_space = observation_space
        self.grid_size = int(grid_size) if grid_size is not None else infer_grid_size(observation_space, state_input_shape[0])
        self.compute_dtype = compute_dtype
        self.overlap_branches = True  # pose branch on a second stream
        o1, o2 = conv_out(self.grid_size)
        self.grid_feat = 16 * o2 ** 3  # 1024 at G=20
        pose_feat = int(state_input_shape[0]) * 4  # 6 -> 24 per pose (sin/cos of x*{1,2})

        self.naive_encoder_grid = nn.Sequential(
            nn.Conv3d(1, 16, kernel_size=3, stride=2, padding=0), nn.BatchNorm3d(16), nn.ReLU(inplace=True),
            nn.Conv3d(16, 16, kernel_size=3, stride=2, padding=0), nn.BatchNorm3d(16), nn.ReLU(inplace=True))
        self.output_layer_grid = nn.Sequential(nn.Linear(self.grid_feat, 256), nn.ReLU(inplace=True))
        self.naive_encoder_action = nn.Sequential(nn.Linear(pose_feat, 256), nn.ReLU(inplace=True),
                                                  nn.Linear(256, 256), nn.ReLU(inplace=True))
        self.output_layer = nn.Sequential(nn.Linear(512, 256), nn.ReLU(inplace=True))
        # 2**arange(2); kept on the module's device (not in the state_dict) so that the forward is
        # capturable in a hipGraph -- the reference re-creates and uploads it on every call (:71)
        self.register_buffer("_freq_bands", 2 ** torch.arange(2).float(), persistent=False)

    @property
    def features_dim(self) -> int:
        return self._features_dim

    def positional_encoding(self, positions: torch.Tensor, freqs: int = 2) -> torch.Tensor:
        """[..., A] -> [..., 4A]: cat(sin(p), cos(p)) of p = (x0*1, x0*2, x1*1, ...)
        (hybrid_encoder.py:63-74)."""
        freq_bands = self._freq_bands if freqs == 2 else (2 ** torch.arange(freqs).float()).to(positions.device)
        pts = (positions[..., None] * freq_bands).reshape(positions.shape[:-1] + (freqs * positions.shape[-1],))
        return torch.cat([torch.sin(pts), torch.cos(pts)], dim=-1)

    def forward(self, observations) -> torch.Tensor:
        from ..ops import encoder_ops
        return encoder_ops.hybrid_forward(self, observations)
