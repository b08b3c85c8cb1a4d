"""Hybrid_Encoder: the multi-source (pose history + occupancy grid) features extractor.

Drop-in for `gennbv/network/hybrid_encoder.py:18-98` behind the SB3 features-extractor
protocol (constructor kwargs of gennbv/train/train_gennbv.py:152-168, `.features_dim`,
`forward(observations[B, D_obs]) -> [B, features_dim]`), with the reference's
`state_dict` key names so its checkpoints load unchanged:

    naive_encoder_grid.{0,1,3,4}.*   Conv3d(1,16,k3,s2) BN Conv3d(16,16,k3,s2) BN
    output_layer_grid.0.*            Linear(16*o2^3, 256)
    naive_encoder_action.{0,2}.*     Linear(stack*24, 256), Linear(256, 256)
    output_layer.0.*                 Linear(512, 256)

Differences, all additive: the grid edge G is a parameter (the reference hard-codes
20^3 / 8000 / 1024, hybrid_encoder.py:47,90-91; G=20 reproduces it exactly), and the
heavy layers can run on the hand-written gfx950 kernels (`backend="hip"`, see
gennbv_amd/ops) instead of torch's library kernels (`backend="torch"`, the fp32
parity reference for the floating-point kernels).
The released encoder never reads the `state_rgb` slice of the observation (:76-98);
neither does this one.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import nn


def conv_out(g: int) -> Tuple[int, int]:
    o1 = (g - 3) // 2 + 1
    return o1, (o1 - 3) // 2 + 1


class Hybrid_Encoder(nn.Module):
    def __init__(self, observation_space, encoder_param=None, net_param=None, visual_input_shape=None,
                 state_input_shape=None, grid_size: int = 20, backend: str = "torch", compute_dtype=torch.float32):
        assert encoder_param is not None, "Need parameters !"
        assert net_param is not None, "Need parameters !"
        assert isinstance(visual_input_shape, (List, Tuple, list, tuple)), "Use tuple or list"
        assert isinstance(state_input_shape, (List, Tuple, list, tuple)), "Use tuple or list"
        super().__init__()
        self.image_channel = visual_input_shape[0]
        self.image_shape = visual_input_shape[1:]
        self.state_input_shape = state_input_shape
        # the reference pops the last entry in place (hybrid_encoder.py:32-33)
        self._features_dim = net_param["append_hidden_shapes"][-1]
        net_param["append_hidden_shapes"].pop()
        self._observation_space = observation_space
        self.grid_size = int(grid_size)
        self.backend = backend
        self.compute_dtype = compute_dtype
        self.overlap_branches = True  # backend="hip": pose branch on a second stream
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
        if self.backend == "hip":
            from ..ops import encoder_ops
            return encoder_ops.hybrid_forward(self, observations)
        if not isinstance(observations, torch.Tensor):  # ops.encoder_ops.RowGather
            observations = observations.materialize()
        num_env = observations.shape[0]
        s = self.state_input_shape[0]
        g = self.grid_size
        action_input = observations[:, :s].view(num_env, -1, 6)
        action_input = self.positional_encoding(action_input).view(num_env, -1)
        grid_input = observations[:, s:s + g ** 3].reshape(num_env, 1, g, g, g)
        feature_action = self.naive_encoder_action(action_input)
        feature_grid = self.naive_encoder_grid(grid_input).reshape(num_env, -1)
        feature_grid = self.output_layer_grid(feature_grid)
        return self.output_layer(torch.cat((feature_action, feature_grid), dim=-1))
