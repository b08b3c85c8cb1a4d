"""Synthetic depth + pose feed that replaces Isaac Gym at the observation boundary.

The reference renders depth / segmentation / RGB with Isaac Gym
(gennbv/env/env_train_gennbv.py:346-354, legged_gym/env/base/*): out of scope
here.  This module produces tensors with exactly the layout and conventions the
hot path consumes (SURVEY.md section 1 "Below the hot path", section 8d):

  depth_raw  [N,H,W] f32   Isaac style: NEGATIVE metres, -inf where the ray hits nothing
  seg_raw    [N,H,W] f32   255 on the object, 0 on the ground plane / miss
  rgba       [N,H,W,4] u8
  view       [N,4,4] f32   Isaac's transposed extrinsics: c2w = inv(view^T) @ blender2opencv
  env_origins[N,3]  f32    per-env world offset subtracted from c2w's translation
  grid_gt    [N,G,G,G] f32 binary surface-occupancy ground truth
  range_gt   [N,6]         (xmax, xmin, ymax, ymin, zmax, zmin) of the voxel centres
  voxel_size [N,3]

Scene = seeded union of 3-8 axis-aligned boxes standing on the ground inside
(+-8, +-8, 0..10) m; camera poses come from the reference action lattice
(gennbv/env/config_gennbv_train.py:62-69); depth by analytic ray/box (slab)
intersection.  Everything is plain torch so it runs on CPU (tests, fixtures)
and on the GPU (bench feed); it is input *generation*, not part of the timed
hot path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

from .config import TaskConfig

BLENDER2OPENCV = torch.tensor([[1.0, 0, 0, 0], [0, -1.0, 0, 0], [0, 0, -1.0, 0], [0, 0, 0, 1.0]])


def camera_intrinsics(h: int, w: int, horizontal_fov_deg: float = 90.0) -> torch.Tensor:
    """Pinhole K as the reference builds it (gennbv/env/env_train_base.py:787-803):
    fx = W/2 / tan(FOVx/2), fy = H/2 / tan(FOVy/2) with FOVy = FOVx * H / W."""
    fov_x = horizontal_fov_deg / 180.0 * math.pi
    fov_y = fov_x * h / w
    fx = 0.5 * w / math.tan(0.5 * fov_x)
    fy = 0.5 * h / math.tan(0.5 * fov_y)
    return torch.tensor([[fx, 0, w / 2], [0, fy, h / 2], [0, 0, 1]]).float()


def inverse_intrinsics(h: int, w: int, horizontal_fov_deg: float = 90.0) -> torch.Tensor:
    """env_train_gennbv.py:168-169: torch.linalg.inv(K) in fp32 (computed on CPU so
    the 9 numbers are identical on every device)."""
    return torch.linalg.inv(camera_intrinsics(h, w, horizontal_fov_deg)).to(torch.float32)


@dataclass
class Scene:
    boxes_min: torch.Tensor  # [N,B,3] (unused boxes have min > max)
    boxes_max: torch.Tensor  # [N,B,3]
    grid_gt: torch.Tensor  # [N,G,G,G] f32
    range_gt: torch.Tensor  # [N,6]
    voxel_size: torch.Tensor  # [N,3]
    num_valid_voxel_gt: torch.Tensor  # [N]
    env_origins: torch.Tensor  # [N,3]


def make_scenes(num_envs: int, grid_size: int, seed: int = 1, device="cpu", max_boxes: int = 8,
                env_spacing: float = 5.0) -> Scene:
    g = torch.Generator(device="cpu").manual_seed(seed)
    n, b = num_envs, max_boxes
    nbox = torch.randint(3, max_boxes + 1, (n,), generator=g)
    centre = (torch.rand(n, b, 2, generator=g) - 0.5) * 9.0  # xy centre in [-4.5, 4.5]
    half = 1.0 + torch.rand(n, b, 2, generator=g) * 2.5  # half extent 1.0..3.5 m
    height = 2.0 + torch.rand(n, b, generator=g) * 7.0  # 2..9 m tall
    bmin = torch.cat([centre - half, torch.zeros(n, b, 1)], -1)
    bmax = torch.cat([centre + half, height[..., None]], -1)
    unused = torch.arange(b)[None, :] >= nbox[:, None]
    bmin[unused] = 1e6
    bmax[unused] = -1e6
    # GT grid: voxel centres at min + i*v, v = range/(G-1) (mirrors env_train_gennbv.py:67-80)
    rng = torch.tensor([8.0, -8.0, 8.0, -8.0, 10.0, 0.0]).repeat(n, 1)
    vox = torch.stack([(rng[:, 0] - rng[:, 1]), (rng[:, 2] - rng[:, 3]), (rng[:, 4] - rng[:, 5])], -1) / (grid_size - 1)
    idx = torch.arange(grid_size, dtype=torch.float32)
    cx = rng[:, 1, None] + idx[None] * vox[:, 0, None]  # [N,G]
    cy = rng[:, 3, None] + idx[None] * vox[:, 1, None]
    cz = rng[:, 5, None] + idx[None] * vox[:, 2, None]
    grid_gt = torch.zeros(n, grid_size, grid_size, grid_size)
    hv = 0.5 * vox  # [N,3]
    for k in range(b):
        lo, hi = bmin[:, k], bmax[:, k]  # [N,3]

        def inside(c, a, grow):
            return (c > (lo[:, a, None] - grow[:, a, None])) & (c < (hi[:, a, None] + grow[:, a, None]))

        outer = (inside(cx, 0, hv)[:, :, None, None] & inside(cy, 1, hv)[:, None, :, None]
                 & inside(cz, 2, hv)[:, None, None, :])
        inner = (inside(cx, 0, -hv)[:, :, None, None] & inside(cy, 1, -hv)[:, None, :, None]
                 & inside(cz, 2, -hv)[:, None, None, :])
        grid_gt = torch.maximum(grid_gt, (outer & ~inner).float())
    side = int(math.ceil(math.sqrt(n)))
    e = torch.arange(n)
    origins = torch.stack([(e % side).float() * env_spacing, (e // side).float() * env_spacing, torch.zeros(n)], -1)
    sc = Scene(bmin, bmax, grid_gt, rng, vox, grid_gt.sum(dim=(1, 2, 3)).clamp(min=1.0), origins)
    return Scene(*[t.to(device) for t in (sc.boxes_min, sc.boxes_max, sc.grid_gt, sc.range_gt, sc.voxel_size,
                                          sc.num_valid_voxel_gt, sc.env_origins)])


def sample_actions(num_envs: int, cfg: TaskConfig, generator: torch.Generator, look_at_scene: bool = True) -> torch.Tensor:
    """Random poses on the reference action lattice, int64 [N,6]."""
    up = torch.tensor(cfg.clip_pose_idx_up)
    a = torch.stack([torch.randint(0, int(u) + 1, (num_envs,), generator=generator) for u in up], -1)
    if look_at_scene:
        # aim the camera at the scene centre (snapped to the yaw / pitch lattice, +-1 step
        # of jitter) so that a large share of the pixels sees the object
        unit = torch.tensor(cfg.action_unit)
        low = torch.tensor(cfg.clip_pose_low)
        a[:, 2] = torch.randint(10, 46, (num_envs,), generator=generator)
        pos = a[:, :3].float() * unit[:3] + low[:3]
        yaw = torch.atan2(-pos[:, 1], -pos[:, 0]) % (2 * math.pi)
        dist = pos[:, :2].norm(dim=-1)
        pitch = torch.atan2(pos[:, 2] - 2.5, dist)
        jit = torch.randint(-1, 2, (num_envs, 2), generator=generator)
        a[:, 5] = (torch.round(yaw / float(unit[5])).long() + jit[:, 0]) % 12
        a[:, 4] = (torch.round((pitch - float(low[4])) / float(unit[4])).long() + jit[:, 1]).clamp(0, 12)
    return a


def poses_from_actions(actions: torch.Tensor, cfg: TaskConfig) -> torch.Tensor:
    """get_pose_from_discrete_action (gennbv/env/env_train_base.py:665-667):
    poses = action * action_unit + clip_pose_low, fp32."""
    unit = torch.tensor(cfg.action_unit, device=actions.device)
    low = torch.tensor(cfg.clip_pose_low, device=actions.device)
    return actions * unit + low


def camera_to_world(poses: torch.Tensor) -> torch.Tensor:
    """OpenCV-convention c2w [N,4,4] (x right, y down, z forward) for a pose
    (x, y, z, roll=0, pitch, yaw): forward = (cos p cos y, cos p sin y, -sin p)."""
    p, y = poses[:, 4].double(), poses[:, 5].double()
    fwd = torch.stack([torch.cos(p) * torch.cos(y), torch.cos(p) * torch.sin(y), -torch.sin(p)], -1)
    right = torch.stack([torch.sin(y), -torch.cos(y), torch.zeros_like(y)], -1)
    down = torch.cross(fwd, right, dim=-1)
    m = torch.zeros(poses.shape[0], 4, 4, dtype=torch.float64, device=poses.device)
    m[:, :3, 0], m[:, :3, 1], m[:, :3, 2] = right, down, fwd
    m[:, :3, 3] = poses[:, :3].double()
    m[:, 3, 3] = 1.0
    return m


def view_matrix_from_c2w(c2w_local: torch.Tensor, env_origins: torch.Tensor) -> torch.Tensor:
    """Isaac-style transposed view matrix V such that the reference's
    inv(V^T) @ blender2opencv, translation - env_origins (env_train_gennbv.py:512-514)
    gives back c2w_local (up to fp32 rounding of the inverse)."""
    m = c2w_local.clone().double()
    m[:, :3, 3] += env_origins.double()
    b2o = BLENDER2OPENCV.double().to(m.device)
    vt = torch.linalg.inv(m @ b2o)  # (c2w_global @ b2o^-1)^-1, b2o is an involution
    return vt.transpose(-2, -1).contiguous().float()


def c2w_from_view(view: torch.Tensor, env_origins: torch.Tensor) -> torch.Tensor:
    """Host-side plumbing of back_projection_fg (env_train_gennbv.py:512-514):
    c2w = inv(view^T) @ blender2opencv ; c2w[:, :3, 3] -= env_origins.
    Evaluated with torch.linalg.inv on the tensor's device like the reference."""
    c2w = torch.linalg.inv(view.transpose(-2, -1)) @ BLENDER2OPENCV.to(view.device).unsqueeze(0)
    c2w[:, :3, 3] -= env_origins
    return c2w.contiguous()


def render_depth(scene: Scene, poses: torch.Tensor, h: int, w: int, horizontal_fov_deg: float = 90.0,
                 with_rgba: bool = True, chunk: int = 16):
    """Analytic ray / box render.  Returns (depth_raw, seg_raw, rgba, view)."""
    dev = poses.device
    n = poses.shape[0]
    kinv = inverse_intrinsics(h, w, horizontal_fov_deg).to(dev)
    us = torch.arange(w, device=dev, dtype=torch.float32)
    vs = torch.arange(h, device=dev, dtype=torch.float32)
    vv, uu = torch.meshgrid(vs, us, indexing="ij")
    pix = torch.stack([uu, vv, torch.ones_like(uu)], -1).view(-1, 3)  # [HW,3]
    dirs_cam = pix @ kinv.T  # [HW,3], z component == 1 -> ray parameter == depth
    c2w = camera_to_world(poses).float()
    depth = torch.empty(n, h * w, device=dev)
    seg = torch.empty(n, h * w, device=dev)
    rgba = torch.zeros(n, h * w, 4, dtype=torch.uint8, device=dev) if with_rgba else None
    inf = float("inf")
    for s in range(0, n, chunk):
        e = slice(s, min(n, s + chunk))
        rot, org = c2w[e, :3, :3], c2w[e, :3, 3]  # [c,3,3], [c,3]
        d = torch.einsum("cij,pj->cpi", rot, dirs_cam)  # [c,HW,3]
        o = org[:, None, :]
        safe = torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
        t_obj = torch.full(d.shape[:2], inf, device=dev)
        which = torch.zeros(d.shape[:2], dtype=torch.long, device=dev)
        for k in range(scene.boxes_min.shape[1]):
            lo = scene.boxes_min[e, k][:, None, :]
            hi = scene.boxes_max[e, k][:, None, :]
            t0 = (lo - o) / safe
            t1 = (hi - o) / safe
            tn = torch.minimum(t0, t1).amax(-1)
            tf = torch.maximum(t0, t1).amin(-1)
            hit = (tf >= tn) & (tf > 1e-3) & (lo[..., 0] < hi[..., 0])
            t = torch.where(tn > 1e-3, tn, tf)
            t = torch.where(hit, t, torch.full_like(t, inf))
            closer = t < t_obj
            which = torch.where(closer, torch.full_like(which, k + 1), which)
            t_obj = torch.minimum(t_obj, t)
        t_gnd = torch.where(d[..., 2] < -1e-6, -o[..., 2] / safe[..., 2], torch.full_like(t_obj, inf))
        t_gnd = torch.where(t_gnd > 1e-3, t_gnd, torch.full_like(t_gnd, inf))
        is_obj = t_obj < t_gnd
        t = torch.minimum(t_obj, t_gnd)
        depth[e] = torch.where(torch.isinf(t), torch.full_like(t, -inf), -t)
        seg[e] = torch.where(is_obj, torch.full_like(t, 255.0), torch.zeros_like(t))
        if with_rgba:
            shade = (which * 29 % 200 + 40).to(torch.uint8)
            rgba[e, :, 0] = torch.where(is_obj, shade, torch.full_like(shade, 90))
            rgba[e, :, 1] = torch.where(is_obj, 255 - shade, torch.full_like(shade, 120))
            rgba[e, :, 2] = torch.where(is_obj, (shade // 2) + 60, torch.full_like(shade, 70))
            rgba[e, :, 3] = 255
    view = view_matrix_from_c2w(camera_to_world(poses), scene.env_origins)
    return (depth.view(n, h, w), seg.view(n, h, w),
            rgba.view(n, h, w, 4) if with_rgba else None, view)


@dataclass
class Frame:
    depth_raw: torch.Tensor
    seg_raw: torch.Tensor
    rgba: Optional[torch.Tensor]
    view: torch.Tensor
    actions: torch.Tensor  # recorded lattice actions int64 [N,6]
    poses: torch.Tensor  # [N,6] f32


def make_frames(scene: Scene, cfg: TaskConfig, num_frames: int, seed: int = 1, with_rgba: bool = True):
    """A recorded feed: `num_frames` frames for every env (poses i.i.d. on the lattice)."""
    g = torch.Generator(device="cpu").manual_seed(seed + 7919)
    dev = scene.grid_gt.device
    n = scene.grid_gt.shape[0]
    frames = []
    for _ in range(num_frames):
        a = sample_actions(n, cfg, g).to(dev)
        poses = poses_from_actions(a, cfg).float()
        d, s, c, v = render_depth(scene, poses, cfg.camera_height, cfg.camera_width, cfg.horizontal_fov, with_rgba)
        frames.append(Frame(d, s, c, v, a, poses))
    return frames
