"""Time the fused voxel update at a BASELINE config (default config 1: 256 x 240x320 x 64^3)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig
from gennbv_amd.env.state_encoding import OccupancyGridUpdater

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--g", type=int, default=64)
ap.add_argument("--h", type=int, default=240)
ap.add_argument("--w", type=int, default=320)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--prob", default="coded", choices=["coded", "f32"], help="1-byte coded probability grid (what ReplayFeedEnv uses) or fp32")
ap.add_argument("--out", default="i8", choices=["f32", "f32+i8", "i8"],
                help="tri-class output: fp32 rows (flat observations), fp32 + int8 copy, or int8 rows only (compact observations, "
                     "coded update; the bench default)")
ap.add_argument("--phase-times", action="store_true", help="experiment builds (-DPHASE_TIMING): per-workgroup phase times of k_hit_list")
ap.add_argument("--no-self-clean", action="store_true", help="coded update with the per-step mask fill (A/B against the self-cleaning masks)")
a = ap.parse_args()
if a.prob != "coded":
    a.out = "f32"
dev = "cuda:0"
cfg = TaskConfig(camera_width=a.w, camera_height=a.h, grid_size=a.g)
scene = S.make_scenes(a.n, a.g, seed=1, device=dev)
frames = S.make_frames(scene, cfg, a.frames, seed=1, with_rgba=False)
upd = OccupancyGridUpdater(a.n, a.g, a.h, a.w, S.inverse_intrinsics(a.h, a.w), scene.range_gt, scene.voxel_size, scene.grid_gt, dev,
                           max_steps_between_resets=100 if a.prob == "coded" else None)
all_reset = torch.ones(a.n, dtype=torch.uint8, device=dev)
t8 = torch.zeros(a.n, a.g ** 3, dtype=torch.int8, device=dev) if "i8" in a.out else None
kw = dict(tri_i8_out=t8, fp32_out=a.out != "i8") if t8 is not None else {}
print("tri-class output:", a.out)
c2ws = [S.c2w_from_view(f.view, scene.env_origins) for f in frames]
poses = [f.poses.contiguous() for f in frames]
upd.self_clean = False  # (the masks of the warm-up steps are read for the statistics line)
for i in range(5):
    upd.update(frames[i % a.frames].depth_raw, frames[i % a.frames].seg_raw, c2ws[i % a.frames], poses[i % a.frames], **kw)
hit, path = upd.masks()
if not a.no_self_clean:
    upd.self_clean, upd._ws_dirty = True, False
    upd.workspace.zero_()  # back to the self-cleaning mode for the timed loop
print("fg frac", float((frames[0].seg_raw > 50).float().mean()), "hit voxels/env", float(hit.flatten(1).sum(1).float().mean()),
      "path voxels/env", float(path.flatten(1).sum(1).float().mean()), "max hit/env", int(hit.flatten(1).sum(1).max()),
      "max path/env", int(path.flatten(1).sum(1).max()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.iters):
    k = i % a.frames
    upd.update(frames[k].depth_raw, frames[k].seg_raw, c2ws[k], poses[k], reset_mask=all_reset if i % 64 == 63 else None, **kw)  # episodes end
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
bvox = a.h * a.w * 8 + a.g ** 3 * 4 * 6 + 200
print(f"update_occ_grid: {ms:.4f} ms/step  -> {a.n / ms * 1e3:.0f} env-steps/s (voxel only), "
      f"algorithmic {a.n * bvox / 1e6:.1f} MB/step -> {a.n * bvox / ms / 1e6:.1f} GB/s = {a.n * bvox / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s")

if a.phase_times:
    import numpy as np
    nrb = ((a.n + 7) // 8) * 8 * 16
    # one more step on a zeroed stamp area, so that every stamp belongs to the SAME launch
    upd.workspace.view(torch.int32)[-(8 * 512 + 8 * nrb):].zero_()
    upd.update(frames[0].depth_raw, frames[0].seg_raw, c2ws[0], poses[0], **kw)
    torch.cuda.synchronize()
    nwg = a.n * max(1, (512 + a.n - 1) // a.n)
    tail = upd.workspace.view(torch.int32)[-8 * nwg:].cpu().numpy().reshape(nwg, 8)
    pa, pb, rays, t0 = tail[:, 0] / 100.0, tail[:, 1] / 100.0, tail[:, 2], tail[:, 3] / 100.0
    print(f"k_hit_list phases over {nwg} workgroups (us): A mean {pa.mean():.1f} max {pa.max():.1f} | B mean {pb.mean():.1f} max {pb.max():.1f} | "
          f"rays mean {rays.mean():.0f} max {rays.max()} | start spread {t0.max() - t0.min():.1f} | end spread {(t0 + pa + pb).max() - t0.min():.1f}")
    i = int(np.argmax(pb))
    print("  B pieces (mask store+popc | scans | emission): mean", (tail[:, 4:7] / 100.0).mean(0).round(1), " heaviest WG:", (tail[i, 4:7] / 100.0).round(1),
          "rays", rays[i], "its A", pa[i])
    print("  corr(A, rays)", np.corrcoef(pa, rays)[0, 1].round(2))

    rt = upd.workspace.view(torch.int32)[-(8 * 512 + 8 * nrb):-8 * 512].cpu().numpy().reshape(nrb, 8)[::-1]
    live = rt[rt[:, 5] == 1]
    t0 = live[:, 0] / 100.0
    print(f"k_ray_list phases over {len(live)} live workgroups of {nrb} (us): wave-0 walk mean {live[:, 1].mean() / 100:.1f} max {live[:, 1].max() / 100:.1f} | "
          f"walk (all waves) mean {live[:, 2].mean() / 100:.1f} max {live[:, 2].max() / 100:.1f} | total mean {live[:, 3].mean() / 100:.1f} max {live[:, 3].max() / 100:.1f} | "
          f"start spread {t0.max() - t0.min():.1f} | end spread {(t0 + live[:, 3] / 100.0).max() - t0.min():.1f}")
