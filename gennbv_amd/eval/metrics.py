"""Metrics of the evaluation env.

* `reconstruction_accuracy_cm`: env_eval_gennbv.py:253-262 -- the episode's back-projected points rounded to 1 cm,
  de-duplicated (torch.unique(dim=0)) and compared with the GT cloud by pytorch3d's chamfer_distance, times 100.
  The Chamfer sum runs on the gfx950 kernel (csrc/chamfer.hip); rounding + unique are torch plumbing.
* `auc_update` / `mean_auc`: evaluation.py:358-378 and :341, vectorised over envs (no per-env Python loop).
"""
from __future__ import annotations

import torch

from .. import _lib

_ws = {}


def chamfer_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """mean_i min_j |x_i - y_j|^2 + mean_j min_i |x_i - y_j|^2 for x [n,3], y [m,3] (fp32, GPU); 0-dim tensor."""
    lib = _lib.load()
    _lib.require_cuda(x, y)
    x, y = x.contiguous().float(), y.contiguous().float()
    assert x.dim() == 2 and x.shape[1] == 3 and y.dim() == 2 and y.shape[1] == 3 and x.shape[0] > 0 and y.shape[0] > 0
    n, m = int(x.shape[0]), int(y.shape[0])
    need = lib.gnbv_chamfer_workspace_bytes(n, m)
    key = str(x.device)
    ws = _ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=x.device)
        _ws[key] = ws
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.check(lib.gnbv_chamfer_distance(x.data_ptr(), n, y.data_ptr(), m, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr(x.device)), "gnbv_chamfer_distance")
    return out[0]


def unique_rounded_points(pts: torch.Tensor, decimals: int = 2) -> torch.Tensor:
    """torch.unique(torch.round(pts, decimals=decimals), dim=0) (env_eval_gennbv.py:256-259): rows in
    lexicographic order.  Implemented on integer keys (rint(p * 10^d) fits 21 bits per axis for |p| < 10^4 m):
    one int64 sort instead of a row-wise lexicographic unique."""
    scale = float(10 ** decimals)
    k = torch.round(pts.float() * scale).to(torch.int64)  # torch.round(x, decimals) == nearbyint(x * 10^d) / 10^d
    lim = 1 << 20
    if pts.numel() == 0 or int(k.abs().max()) >= lim:
        return torch.unique(torch.round(pts, decimals=decimals), dim=0)
    key = ((k[:, 0] + lim) << 42) | ((k[:, 1] + lim) << 21) | (k[:, 2] + lim)
    key = torch.unique(key)  # sorted: lexicographic in (x, y, z) like unique(dim=0)
    out = torch.stack((((key >> 42) & 0x1FFFFF) - lim, ((key >> 21) & 0x1FFFFF) - lim, (key & 0x1FFFFF) - lim), dim=1)
    return out.to(torch.float32) / scale


def reconstruction_accuracy_cm(scanned_pts: torch.Tensor, pc_gt: torch.Tensor) -> torch.Tensor:
    """accuracy = chamfer(unique(round(pts, 2)), pc_gt) * 100   (env_eval_gennbv.py:255-261)."""
    return chamfer_distance(unique_rounded_points(scanned_pts, 2), pc_gt) * 100.0


def auc_update(auc_rews: torch.Tensor, cur_rewards: torch.Tensor, cur_length: int, dones: torch.Tensor,
               episode_done_flag: torch.Tensor) -> torch.Tensor:
    """evaluation.py:358-378: column cur_length-1 of the [n_envs, max_length] curve = the reward of this step,
    or -- for envs whose episode already ended -- the previous column; envs that end on THIS step keep the
    column's old value."""
    c = cur_length - 1
    ended = episode_done_flag.to(torch.bool).to(auc_rews.device)
    running = (~ended) & (dones.to(auc_rews.device) == 0)
    col = auc_rews[:, c].clone()
    col = torch.where(running, cur_rewards.to(auc_rews), col)
    col = torch.where(ended, auc_rews[:, c - 1], col)  # c - 1 == -1 wraps to the last column exactly like the reference's index
    auc_rews[:, c] = col
    return auc_rews


def mean_auc(auc_rews: torch.Tensor) -> torch.Tensor:
    """evaluation.py:341: sum_idx AUC[:, idx] * (max_length - idx) / max_length  -> [n_envs]."""
    max_length = auc_rews.shape[1]
    w = (max_length - torch.arange(max_length, device=auc_rews.device, dtype=auc_rews.dtype))
    return (auc_rews * w).sum(1) / max_length
