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
    nrb = ((a.n + 7) // 8) * 8 * 16  # (stamp slots: the ray launch's grid is at most this)
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
    live, dead = rt[rt[:, 5] == 1], rt[rt[:, 5] == 2]
    t0 = live[:, 0] / 100.0
    tall = np.concatenate([live[:, 0], dead[:, 0]]) / 100.0
    print(f"k_ray_list: {len(dead)} dead workgroups, entry -> exit mean {dead[:, 6].mean() / 100:.2f} max {dead[:, 6].max() / 100:.2f} us; all entries spread {tall.max() - tall.min():.1f}; "
          f"live: count known mean {live[:, 6].mean() / 100:.2f} max {live[:, 6].max() / 100:.2f}, pose known + mask clear mean {live[:, 7].mean() / 100:.2f} max {live[:, 7].max() / 100:.2f}")
    order = np.argsort(t0)
    print("  live starts (us after the first), every 10th percentile:", np.percentile(t0 - tall.min(), range(0, 101, 10)).round(1))
    print("  live ends, every 10th percentile:", np.percentile(t0 + live[:, 3] / 100.0 - tall.min(), range(0, 101, 10)).round(1))
    print(f"k_ray_list phases over {len(live)} live workgroups of {nrb} (us): wave-0 walk mean {live[:, 1].mean() / 100:.1f} max {live[:, 1].max() / 100:.1f} | "
          f"total mean {live[:, 3].mean() / 100:.1f} max {live[:, 3].max() / 100:.1f} | "
          f"start spread {t0.max() - t0.min():.1f} | end spread {(t0 + live[:, 3] / 100.0).max() - t0.min():.1f}")
    lv_idx = np.nonzero(rt[:, 5] == 1)[0]
    dur = rt[lv_idx, 3] / 100.0
    print("  live duration percentiles (us):", np.percentile(dur, [10, 50, 90, 99, 100]).round(1))
    # where the workgroups ran: HW_ID (cu_id [11:8], sh_id [12], se_id [15:13]) and XCC_ID [3:0]
    def place(r, hw_col, xcc_col):
        hw, xcc = r[:, hw_col], r[:, xcc_col] & 15
        return xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15
    blk = np.arange(nrb)[::1]
    rec = rt  # (row i = block i)
    has = rec[:, 5] > 0
    xcc = np.where(rec[:, 5] == 1, rec[:, 4], rec[:, 1]) & 15
    hw = rec[:, 2]
    se, sh, cu = (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15
    b = np.arange(nrb)
    print("  block -> XCC == block % 8 for", int(((xcc == (b & 7)) & has).sum()), "of", int(has.sum()), "; distinct (se, sh, cu) per XCC:",
          len(set(zip(xcc[has], se[has], sh[has], cu[has]))) / max(1, len(set(xcc[has]))))
    slot = b >> 3
    for k in range(16):
        m = has & (xcc == 0) & (slot < 16 * 3)
    first = [(int(slot[i]), int(se[i]), int(sh[i]), int(cu[i]), int(rec[i, 5])) for i in np.nonzero(has & (xcc == 0))[0][:48]]
    print("  XCC 0, first 48 blocks (slot, se, sh, cu, live=1/dead=2):", first)
    lv = rec[:, 5] == 1
    for x in range(2):
        mm = lv & (xcc == x)
        key = se[mm] * 100 + sh[mm] * 16 + cu[mm]
        u, c = np.unique(key, return_counts=True)
        print(f"  XCC {x}: live workgroups per CU: n_cu {len(u)} min {c.min()} max {c.max()} mean {c.mean():.1f}; per SE:", {int(k): int(((se[mm]) == k).sum()) for k in np.unique(se[mm])})
