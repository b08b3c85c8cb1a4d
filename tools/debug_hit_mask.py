"""GPU debug: hit mask of the library (GENNBV_HIP_LIB) against per-pixel voxels from the CPU oracle; prints the pixels behind missing / extra voxels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig
from gennbv_amd.env.state_encoding import OccupancyGridUpdater
from oracle import oracle as orc
n, h, w, g = int(os.environ.get("N", 8)), 240, 320, 64
dev = "cuda:0"
cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
scene = S.make_scenes(n, g, seed=1, device=dev)
f = S.make_frames(scene, cfg, 1, seed=1, with_rgba=False)[0]
upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, dev)
upd.self_clean = False
c2w = S.c2w_from_view(f.view, scene.env_origins)
upd.update(f.depth_raw, f.seg_raw, c2w, f.poses.contiguous())
hit, path = upd.masks()
hit = hit.cpu().numpy().reshape(n, -1)
dp, sp = orc.post_process_depth(f.depth_raw.cpu().numpy(), f.seg_raw.cpu().numpy())
world, fg = orc.back_projection(dp, sp, c2w.cpu().numpy(), S.inverse_intrinsics(h, w).numpy())
idx = orc.points_to_idx(world, fg, scene.range_gt.cpu().numpy(), scene.voxel_size.cpu().numpy(), g)
print("idx", idx.shape, idx.dtype, "fg", fg.shape)
idx = idx.reshape(n, h * w, -1)
for e in range(n):
    v = idx[e]
    if v.shape[-1] == 3:
        keep = (v[:, 0] >= 0)
        lin = (v[:, 0] * g + v[:, 1]) * g + v[:, 2]
    else:
        keep = v[:, 0] >= 0; lin = v[:, 0]
    ref = np.zeros(g ** 3, bool); ref[lin[keep]] = True
    miss = np.nonzero(ref & ~hit[e])[0]; extra = np.nonzero(~ref & hit[e])[0]
    print("env", e, "ref", ref.sum(), "hit", hit[e].sum(), "missing", len(miss), "extra", len(extra))
    for m in miss[:6]:
        px = np.nonzero(keep & (lin == m))[0]
        print("   voxel", m, "pixels", [(int(p), int(p) // 4096, (int(p) % 4096) // 4, int(p) % 4) for p in px[:8]], "(p, tile, lane, k)")
