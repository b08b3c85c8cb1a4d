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
                 state_input_shape=None, grid_size: Optional[int] = None, compute_dtype=torch.float32, semantic_branch: bool = False):
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
        if compute_dtype not in (None, torch.float32):
            # (the bf16 activation-storage mode of rounds 1-3 is gone: slower than the fp32-accurate split-f16 kernels and 1e-3-class loss
            # deltas -- DESIGN.md section 5, "Precision")
            raise ValueError("Hybrid_Encoder computes in fp32 (fp32-accurate products on the f16 matrix pipe); compute_dtype must be torch.float32")
        self.compute_dtype = torch.float32
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
        # Semantic branch (BASELINE configs[2]) -- OPT-IN and BUILD-DEFINED: the released reference renders the two
        # 64 x 64 gray frames into obs["state_rgb"] (env_train_base.py:517-520, env_train_gennbv.py:359-366) and its encoder never reads
        # them (hybrid_encoder.py:76-98), so there is nothing to be equal to ("parity unpinned"); default off, and then the module
        # tree / state_dict is the reference's.  On: the frames (x 1/255) are cut into 8 x 8 patches, every patch of both frames goes
        # through Linear(128, 64) + ReLU (a strided Conv2d(2, 64, k8, s8) written as a GEMM), the 64 patch embeddings through
        # Linear(4096, 256) + ReLU, and the 256 features join the pose and grid features in front of output_layer (768 inputs) --
        # all on the split-K linear kernels of fc_grid (csrc/linear.hip).
        self.semantic_branch = bool(semantic_branch)
        if self.semantic_branch:
            self.naive_encoder_rgb = nn.Sequential(nn.Linear(2 * 8 * 8, 64), nn.ReLU(inplace=True))
            self.output_layer_rgb = nn.Sequential(nn.Linear(64 * 64, 256), nn.ReLU(inplace=True))
        self.output_layer = nn.Sequential(nn.Linear(768 if self.semantic_branch else 512, 256), nn.ReLU(inplace=True))
        # 2**arange(2); kept on the module's device (not in the state_dict) so that the forward is
        # capturable in a hipGraph -- the reference re-creates and uploads it on every call (:71)
        self.register_buffer("_freq_bands", 2 ** torch.arange(2).float(), persistent=False)
        # operand ranges of the split-f16 kernels (check_operand_ranges): a device word the kernels OR activation-bound bits into
        # (GnbvEncoderParams.range_flag) and the switch that keeps this encoder on the fp32-MFMA kernels (.force_fp32)
        self.register_buffer("_range_flag", torch.zeros(1, dtype=torch.int32), persistent=False)
        self.force_fp32 = False

    @property
    def features_dim(self) -> int:
        return self._features_dim

    def positional_encoding(self, positions: torch.Tensor, freqs: int = 2) -> torch.Tensor:
        """[..., A] -> [..., 4A]: cat(sin(p), cos(p)) of p = (x0*1, x0*2, x1*1, ...)
        (hybrid_encoder.py:63-74)."""
        freq_bands = self._freq_bands if freqs == 2 else (2 ** torch.arange(freqs).float()).to(positions.device)
        pts = (positions[..., None] * freq_bands).reshape(positions.shape[:-1] + (freqs * positions.shape[-1],))
        return torch.cat([torch.sin(pts), torch.cos(pts)], dim=-1)

    # Limits of the split-f16 arithmetic (csrc/conv_split.h, csrc/linear.hip): operands are written as f16 hi + lo
    # halves under fixed power-of-two scalings and CLAMPED beyond them.  The margins below leave room for one train() call's
    # worth of Adam steps (|delta w| <= lr per step) and for batch statistics that differ from the running ones.
    W2_LIMIT, W_FC_LIMIT, Z1_LIMIT = 62.0, 15.5, 126.0

    def check_operand_ranges(self, raise_on_flag: bool = True) -> dict:
        """Make the range limits of the split-f16 kernels LOUD instead of silent.  Two parts, one host read each:

        * parameter pre-check: max |W2| (limit 63.4), max |W| of fc_grid and the pose linears (15.8), and BatchNorm-1's activation
          bound |scale| sum|W1| + |scale b1 + shift| from the RUNNING statistics (clamp at 253.9, checked at half of it).  A
          violation switches this encoder to the fp32-MFMA kernels (`force_fp32`; exact, slower) -- nothing was computed yet;
        * activation flags the kernels raised since the last call (bit 2: BatchNorm-1's worst-case bound from the BATCH statistics -- a
          CONSERVATIVE bound: a minibatch of near-constant grids has a tiny variance, hence a large scale, with nothing near the clamp;
          bit 4: a feature above 1000, from the real values): the calls in between MAY have clamped -> `force_fp32` is set for what
          follows and, with `raise_on_flag`, GennbvHipError is raised.  `raise_on_flag=False` returns the flag in `info["flag"]` for a
          caller that can repeat the work on the fp32 kernels (PPO_Grid_Obs.train() snapshots its state and replays the call).

        Called by PPO_Grid_Obs at the start and end of train() and collect_rollouts(); cheap (a few reductions, 2 reads)."""
        from .. import _lib
        seq = self.naive_encoder_grid
        info = {"force_fp32": bool(self.force_fp32), "flag": 0}
        if not seq[0].weight.is_cuda:
            return info
        with torch.no_grad():
            w1, b1, bn1 = seq[0].weight.reshape(16, -1), seq[0].bias, seq[1]
            sc = bn1.weight * torch.rsqrt(bn1.running_var + bn1.eps)
            z1 = sc.abs() * w1.abs().sum(1) + (sc * b1 + bn1.bias - bn1.running_mean * sc).abs()
            lin = [self.output_layer_grid[0]] + [m for m in self.naive_encoder_action if isinstance(m, nn.Linear)]
            if self.semantic_branch:
                lin += [self.naive_encoder_rgb[0], self.output_layer_rgb[0]]
            # the pose branch's second linear reads relu(W p + b) with |p| <= 1 (sin / cos): bounded by its rows' absolute sums
            pose = [m for m in self.naive_encoder_action if isinstance(m, nn.Linear)]
            xpose = (pose[0].weight.abs().sum(1) + pose[0].bias.abs()).max() if len(pose) > 1 else torch.zeros((), device=w1.device)
            if self.semantic_branch:  # (same bound for the patch embeddings: inputs in 0 .. 1)
                xpose = torch.maximum(xpose, (self.naive_encoder_rgb[0].weight.abs().sum(1) + self.naive_encoder_rgb[0].bias.abs()).max())
            vals = torch.stack([seq[3].weight.abs().max(), torch.stack([m.weight.abs().max() for m in lin]).max(), z1.max(), xpose,
                                self._range_flag[0].float()]).cpu()
        w2max, wfcmax, z1max, xpmax, flag = (float(v) for v in vals)
        info.update(w2_max=w2max, w_fc_max=wfcmax, z1_bound=z1max, pose_hidden_bound=xpmax, flag=int(flag))
        bad = not (w2max < self.W2_LIMIT and wfcmax < self.W_FC_LIMIT and z1max < self.Z1_LIMIT and xpmax < 1000.0)  # (a NaN fails too)
        if (bad or flag) and not self.force_fp32:
            self.force_fp32 = True
            for m in lin:
                m._fp32_arith = True
            info["force_fp32"] = True
        if flag:
            self._range_flag.zero_()
        if flag and raise_on_flag:
            raise _lib.GennbvHipError(
                f"split-f16 kernels: an activation left its range (flag {int(flag)}: 2 = relu(bn1(conv1)) bound above 253, 4 = a feature "
                "above 1000); results since the last check may be clamped.  This encoder now runs on the fp32-MFMA kernels "
                "(force_fp32): repeat the call.")
        return info

    @staticmethod
    def rgb_patches(rgb: torch.Tensor) -> torch.Tensor:
        """[B, 2 * 64 * 64] gray frames (values 0 .. 255) -> [B * 64, 128]: patch (py, px) of both frames, scaled to 0 .. 1."""
        b = rgb.shape[0]
        x = rgb.reshape(b, 2, 8, 8, 8, 8).permute(0, 2, 4, 1, 3, 5)  # (b, c, py, iy, px, ix) -> (b, py, px, c, iy, ix)
        return (x * (1.0 / 255.0)).reshape(b * 64, 128)

    def forward(self, observations) -> torch.Tensor:
        from ..ops import encoder_ops
        return encoder_ops.hybrid_forward(self, observations)
