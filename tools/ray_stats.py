"""Ray statistics of the voxel update's workload (GPU): rays per env, ray lengths, how much a wave loses to its longest ray.

    python tools/ray_stats.py [--n 256 --g 64]

The numbers size k_ray_list (csrc/voxel.hip): its cost is wave-steps = sum over waves of the longest ray of the wave.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig
from gennbv_amd.env.state_encoding import OccupancyGridUpdater

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--g", type=int, default=64)
ap.add_argument("--h", type=int, default=240)
ap.add_argument("--w", type=int, default=320)
a = ap.parse_args()
n, g, h, w = a.n, a.g, a.h, a.w
dev = "cuda:0"
cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
scene = S.make_scenes(n, g, seed=1, device=dev)
frames = S.make_frames(scene, cfg, 4, seed=1, with_rgba=False)
upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, dev, max_steps_between_resets=100)
t8 = torch.zeros(n, g ** 3, dtype=torch.int8, device=dev)
upd.self_clean = False
rng = np.random.default_rng(0)
for k, f in enumerate(frames):
    upd.update(f.depth_raw, f.seg_raw, S.c2w_from_view(f.view, scene.env_origins), f.poses.contiguous(), tri_i8_out=t8, fp32_out=False)
    hit, path = upd.masks()
    hit = hit.reshape(n, g, g, g).cpu().numpy().astype(bool)
    path = path.reshape(n, -1).cpu().numpy().astype(bool)
    pose = f.poses[:, :3].cpu().numpy()
    rg, vs = scene.range_gt.cpu().numpy(), scene.voxel_size.cpu().numpy()
    src = np.floor((pose - (rg[:, 1::2] - 0.5 * vs)) / vs).astype(np.int64)
    Ls, cnts, rnd, srt, perfect, words = [], [], 0, 0, 0, []
    for e in range(n):
        t = np.argwhere(hit[e])
        cnts.append(len(t))
        words.append(int(path[e].reshape(-1, 32).any(1).sum()))
        if len(t) == 0:
            continue
        L = np.abs(t - src[e]).max(1)  # steps of the walk (the source voxel itself is set once per workgroup)
        Ls.append(L)
        Lp = rng.permutation(L)
        for i in range(0, len(Lp), 64):
            rnd += Lp[i:i + 64].max()
        Lq = np.sort(L)[::-1]
        for i in range(0, len(Lq), 64):
            srt += Lq[i]
        perfect += L.sum() / 64
    A, cnts, words = np.concatenate(Ls), np.array(cnts), np.array(words)
    print(f"frame {k}: rays/env mean {cnts.mean():.0f} median {np.median(cnts):.0f} max {cnts.max()} empty envs {(cnts == 0).sum()}; "
          f"256-ray slices {np.ceil(cnts / 256).sum():.0f}, 512-ray {np.ceil(cnts / 512).sum():.0f}, 1024-ray {np.ceil(cnts / 1024).sum():.0f}")
    print(f"   steps per ray mean {A.mean():.1f} p50 {np.median(A):.0f} p90 {np.percentile(A, 90):.0f} max {A.max()}; hist/8 {np.histogram(A, bins=range(0, 73, 8))[0]}")
    print(f"   wave-steps: random lanes {rnd}, length-sorted per env {srt}, perfect {perfect:.0f}")
    print(f"   non-zero path words per env: mean {words.mean():.0f} max {words.max()} (of {g ** 3 // 32})")
