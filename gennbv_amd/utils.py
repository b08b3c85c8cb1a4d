"""Drop-in counterparts of the operators in the reference's `gennbv/utils.py`,
running on hand-written gfx950 kernels through the C-ABI (include/gennbv_hip.h).

Same names, argument meaning and error behaviour as the reference:

    bresenham3D_pycuda      gennbv/utils.py:24-227   (alias of bresenham3D_hip)
    scanned_pts_to_idx_3D   gennbv/utils.py:230-270
    pose_coord_to_idx_3D    gennbv/utils.py:273-306
    grid_occupancy_tri_cls  gennbv/utils.py:309-325

plus `post_process_depth` / `back_projection_fg` for the two env methods that
feed them (gennbv/env/env_train_base.py:521-534, env_train_gennbv.py:494-533).
These are the function-granular operators (used by the parity tests and by a
maintainer who swaps one call at a time); the training path uses the fused
`gennbv_amd.env.state_encoding.OccupancyGridUpdater`, which does the same work
in three launches per step for the whole batch.
"""
from __future__ import annotations

from typing import List, Sequence, Union

import torch

from . import _lib


def _st(t):
    return _lib.stream_ptr(t.device)


def bresenham3D_hip(pts_source: torch.Tensor, pts_target: torch.Tensor, map_size) -> torch.Tensor:
    """3-D Bresenham from ONE source voxel to every target voxel.

    pts_source [1,3] (any integer/float dtype, truncated like `.int()`), pts_target
    [num_rays,3]; returns the concatenated in-bounds voxels of all rays, ray order,
    int64 [M,3] -- exactly the reference's return value (utils.py:223-227)."""
    if isinstance(map_size, (list, tuple)):
        assert len(map_size) == 3 and map_size[0] == map_size[1] == map_size[2], "map_size must be cubic"
        map_size = map_size[0]
    map_size = int(map_size)
    _lib.require_cuda(pts_source, pts_target)
    lib = _lib.load()
    source_pts = pts_source.int().contiguous()
    target_pts = pts_target.int().contiguous()
    num_rays = target_pts.shape[0]
    max_pts_per_ray = map_size * 3
    device = pts_source.device
    trajectory_pts = torch.zeros((num_rays, max_pts_per_ray, 3), dtype=torch.int32, device=device)
    trajectory_lengths = torch.zeros(num_rays, dtype=torch.int32, device=device)
    _lib.check(lib.gnbv_bresenham3d(source_pts.data_ptr(), target_pts.data_ptr(), num_rays, map_size,
                                    trajectory_pts.data_ptr(), trajectory_lengths.data_ptr(), _st(pts_source)),
               "gnbv_bresenham3d")
    mask = torch.arange(max_pts_per_ray, device=device)[None, :] < trajectory_lengths[:, None]
    results = trajectory_pts[mask.unsqueeze(-1).expand(-1, -1, 3)].view(-1, 3)
    return results.to(torch.long)


bresenham3D_pycuda = bresenham3D_hip  # the reference's name, for one-line swaps


def bresenham3D_raw(pts_source, pts_target, map_size):
    """Raw kernel outputs (trajectory_pts [R,3G,3] int32, trajectory_lengths [R] int32)."""
    _lib.require_cuda(pts_source, pts_target)
    lib = _lib.load()
    s = pts_source.int().contiguous()
    t = pts_target.int().contiguous().view(-1, 3)
    traj = torch.zeros((t.shape[0], 3 * map_size, 3), dtype=torch.int32, device=t.device)
    lens = torch.zeros(t.shape[0], dtype=torch.int32, device=t.device)
    _lib.check(lib.gnbv_bresenham3d(s.data_ptr(), t.data_ptr(), t.shape[0], int(map_size), traj.data_ptr(),
                                    lens.data_ptr(), _st(t)), "gnbv_bresenham3d")
    return traj, lens


def points_to_idx(world: torch.Tensor, fg: torch.Tensor, range_gt: torch.Tensor, voxel_size_gt: torch.Tensor,
                  map_size: int) -> torch.Tensor:
    """Per-point voxel indices [N,P,3] int32, (-1,-1,-1) where the point is dropped."""
    _lib.require_cuda(world, fg, range_gt, voxel_size_gt)
    lib = _lib.load()
    world = world.contiguous().float()
    fg8 = fg.to(torch.uint8).contiguous()
    n, p = world.shape[0], world.shape[1]
    idx = torch.empty((n, p, 3), dtype=torch.int32, device=world.device)
    _lib.check(lib.gnbv_points_to_idx(world.data_ptr(), fg8.data_ptr(), range_gt.contiguous().float().data_ptr(),
                                      voxel_size_gt.contiguous().float().data_ptr(), n, p, int(map_size),
                                      idx.data_ptr(), _st(world)), "gnbv_points_to_idx")
    return idx


def scanned_pts_to_idx_3D(pts_target: Sequence[torch.Tensor], range_gt: torch.Tensor, voxel_size_gt: torch.Tensor,
                          map_size: int = 256) -> List[Union[torch.Tensor, list]]:
    """List (one per env) of unique, clamped voxel indices int64 [m_i,3], or [] for an
    env without in-bound points -- the reference's contract (utils.py:258-268).
    Rows are in lexicographic order like torch.unique(dim=0)."""
    out: List[Union[torch.Tensor, list]] = []
    g = int(map_size)
    for env_idx, pts in enumerate(pts_target):
        if len(pts) == 0:
            out.append([])
            continue
        ones = torch.ones(1, pts.shape[0], dtype=torch.uint8, device=pts.device)
        idx = points_to_idx(pts.view(1, -1, 3), ones, range_gt[env_idx:env_idx + 1], voxel_size_gt[env_idx:env_idx + 1], g)[0]
        keep = idx[:, 0] >= 0
        if not bool(keep.any()):
            out.append([])
            continue
        lin = (idx[keep, 0].long() * g + idx[keep, 1].long()) * g + idx[keep, 2].long()
        lin = torch.unique(lin)  # sorted 1-D unique == lexicographic row order
        out.append(torch.stack([lin // (g * g), (lin // g) % g, lin % g], dim=-1))
    return out


def pose_coord_to_idx_3D(poses: torch.Tensor, range_gt: torch.Tensor, voxel_size_gt: torch.Tensor, map_size: int = 256,
                         if_col: bool = False) -> torch.Tensor:
    """Camera voxel per env, int64 [N,3], NOT clamped (utils.py:297-306)."""
    assert poses.shape[1] == 3, f"Invalid poses shape: {poses.shape}"
    _lib.require_cuda(poses, range_gt, voxel_size_gt)
    lib = _lib.load()
    p = poses.contiguous().float()
    out = torch.empty((p.shape[0], 3), dtype=torch.int64, device=p.device)
    _lib.check(lib.gnbv_pose_to_idx(p.data_ptr(), range_gt.contiguous().float().data_ptr(),
                                    voxel_size_gt.contiguous().float().data_ptr(), p.shape[0], out.data_ptr(), _st(p)),
               "gnbv_pose_to_idx")
    if if_col:
        out[(out < 0).any(dim=-1)] = -1
        out[(out > map_size - 1).any(dim=-1)] = -1
    return out


def grid_occupancy_tri_cls(grid_prob: torch.Tensor, threshold_occu: float = 0.5, threshold_free: float = 0.0,
                           return_tri_cls_only: bool = False):
    _lib.require_cuda(grid_prob)
    lib = _lib.load()
    gp = grid_prob.contiguous().float()
    tri = torch.empty_like(gp)
    _lib.check(lib.gnbv_grid_tri_cls(gp.data_ptr(), gp.numel(), float(threshold_occu), float(threshold_free),
                                     tri.data_ptr(), _st(gp)), "gnbv_grid_tri_cls")
    if return_tri_cls_only:
        return tri
    return (gp > threshold_occu).to(torch.float32), tri


def post_process_depth(depth_raw: torch.Tensor, seg_raw: torch.Tensor, depth_sense_dist: float = -50.0):
    """Depth / seg branch of post_process_camera_tensor (env_train_base.py:521-534)."""
    _lib.require_cuda(depth_raw, seg_raw)
    lib = _lib.load()
    d, s = depth_raw.contiguous().float(), seg_raw.contiguous().float()
    do, so = torch.empty_like(d), torch.empty_like(s)
    _lib.check(lib.gnbv_post_process_depth(d.data_ptr(), s.data_ptr(), d.numel(), float(depth_sense_dist),
                                           do.data_ptr(), so.data_ptr(), _st(d)), "gnbv_post_process_depth")
    return do, so


def rgb_to_gray(rgba: torch.Tensor, out_h: int = 64, out_w: int = 64, out: torch.Tensor = None,
                out_row_stride: int = None) -> torch.Tensor:
    """RGBA u8 [N,H,W,4] -> nearest resize -> grayscale f32 [N,1,oh,ow]
    (env_train_base.py:517-520; torchvision branch, parity unpinned)."""
    _lib.require_cuda(rgba)
    lib = _lib.load()
    r = rgba.contiguous()
    assert r.dtype == torch.uint8 and r.shape[-1] == 4
    n, h, w, _ = r.shape
    if out is None:
        out = torch.empty((n, 1, out_h, out_w), dtype=torch.float32, device=r.device)
        out_row_stride = out_h * out_w
    _lib.check(lib.gnbv_rgb_to_gray(r.data_ptr(), n, h, w, out_h, out_w, out.data_ptr(), int(out_row_stride), _st(r)),
               "gnbv_rgb_to_gray")
    return out


def back_projection_fg(depth_processed: torch.Tensor, seg_processed: torch.Tensor, c2w: torch.Tensor,
                       inv_intri: torch.Tensor, return_all: bool = False):
    """Env_Train_GenNBV.back_projection_fg (env_train_gennbv.py:494-533) for given c2w.

    Returns the reference's list of per-env foreground world points [n_i,3]; with
    `return_all` also the dense (world [N,HW,3], fg [N,HW] bool) pair."""
    _lib.require_cuda(depth_processed, seg_processed, c2w)
    lib = _lib.load()
    d, s = depth_processed.contiguous().float(), seg_processed.contiguous().float()
    n, h, w = d.shape
    ki = inv_intri.detach().to("cpu", torch.float32).contiguous()  # [host] 9 floats
    world = torch.empty((n, h * w, 3), dtype=torch.float32, device=d.device)
    fg = torch.empty((n, h * w), dtype=torch.uint8, device=d.device)
    _lib.check(lib.gnbv_back_projection(d.data_ptr(), s.data_ptr(), c2w.contiguous().float().data_ptr(), ki.data_ptr(),
                                        n, h, w, world.data_ptr(), fg.data_ptr(), _st(d)), "gnbv_back_projection")
    fgb = fg.bool()
    pts = [world[i][fgb[i]] for i in range(n)]
    return (pts, world, fgb) if return_all else pts
