"""LDS-atomic conflict statistics of k_ray_list's walk (csrc/voxel.hip::walk_packed) for different ORDERS of an env's ray list, simulated
on the CPU from the hit masks of the synthetic workload (GPU needed for the masks only).

    python tools/ray_conflicts.py [--envs 32]

Model: a wave = 64 consecutive list entries, all lanes step together; a lane whose voxel equals its left neighbour's leaves the
atomic to it; ds_or_b32 is serviced in two groups of 32 lanes, 32 banks of 4 bytes, and lanes on one bank (same address or not)
serialise: cost of a wave-step = sum over the two groups of the busiest bank's lane count."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig
from gennbv_amd.env.state_encoding import OccupancyGridUpdater

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32)
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--g", type=int, default=64)
a = ap.parse_args()
n, g, h, w = a.n, a.g, 240, 320
dev = "cuda:0"
cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
scene = S.make_scenes(n, g, seed=1, device=dev)
frames = S.make_frames(scene, cfg, 2, seed=1, with_rgba=False)
upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, dev, max_steps_between_resets=100)
t8 = torch.zeros(n, g ** 3, dtype=torch.int8, device=dev)
upd.self_clean = False
f = frames[1]
upd.update(f.depth_raw, f.seg_raw, S.c2w_from_view(f.view, scene.env_origins), f.poses.contiguous(), tri_i8_out=t8, fp32_out=False)
hit, _ = upd.masks()
hit = hit.reshape(n, -1).cpu().numpy().astype(bool)
pose = f.poses[:, :3].cpu().numpy()
rg, vs = scene.range_gt.cpu().numpy(), scene.voxel_size.cpu().numpy()
src = np.floor((pose - (rg[:, 1::2] - 0.5 * vs)) / vs).astype(np.int64)


def orders(lin):
    """lin: ascending voxel indices of one env -> dict of list orders"""
    out = {"z-fastest (now)": lin}
    wi, bit = lin >> 5, lin & 31
    # plane-major inside groups of 64 non-zero words (k_hit_list: one lane per non-zero word, one wave per 64 of them)
    uw = np.unique(wi)
    grp = np.searchsorted(uw, wi) // 64
    out["bit-plane-major per 64 words"] = lin[np.lexsort((wi, bit, grp))]
    x, y, z = lin // (g * g), (lin // g) % g, lin % g
    out["z-slowest (z, x, y)"] = lin[np.lexsort((y, x, z))]
    out["y-fastest (x, z, y)"] = lin[np.lexsort((y, z, x))]
    return out


def walk_cost(lin, s, dedupe="left"):
    t = np.stack([lin // (g * g), (lin // g) % g, lin % g], 1)
    d = np.abs(t - s)
    da = d.max(1)
    ax = np.where(da == d[:, 0], 0, np.where(da == d[:, 1], 1, 2))
    sg = np.where(t > s, 1, -1)
    cost = steps = lanes_act = 0
    pad = (-len(lin)) % 64
    daw = np.concatenate([da, np.zeros(pad, int)]).reshape(-1, 64)
    for j in range(1, int(da.max()) + 1):
        # closed form of the reference's Bresenham: dominant axis a0 + s j, minors b0 + s floor((2 db j + da) / (2 da))
        pos = np.empty_like(t)
        for k in range(3):
            nb = (2 * d[:, k] * j + da) // np.maximum(2 * da, 1)
            pos[:, k] = s[k] + sg[:, k] * np.where(ax == k, j, nb)
        l = (pos[:, 0] * g + pos[:, 1]) * g + pos[:, 2]
        lw = np.concatenate([l, np.full(pad, -1)]).reshape(-1, 64)
        act = daw >= j
        lw = np.where(act, lw, -1)
        left = np.concatenate([np.full((lw.shape[0], 1), -1), lw[:, :-1]], 1)
        if dedupe == "left":
            do = act & (lw != left)
        else:  # ideal: one lane per distinct voxel of the wave
            srt = np.sort(lw, 1)
            first = np.concatenate([np.ones((lw.shape[0], 1), bool), srt[:, 1:] != srt[:, :-1]], 1) & (srt >= 0)
            lw, do = srt, first
        bank = (lw >> 5) & 31
        c = np.zeros(lw.shape[0], int)
        for grp in range(2):
            bb = np.where(do[:, 32 * grp:32 * grp + 32], bank[:, 32 * grp:32 * grp + 32], -1)
            cnt = np.zeros((lw.shape[0], 33), int)
            np.add.at(cnt, (np.repeat(np.arange(lw.shape[0]), 32), (bb + 1).ravel()), 1)
            c += cnt[:, 1:].max(1)
        wave_act = act.any(1)
        cost += c[wave_act].sum()
        steps += wave_act.sum()
        lanes_act += do.sum()
    return cost, steps, lanes_act


tot = {}
for e in range(a.envs):
    lin = np.nonzero(hit[e])[0]
    if len(lin) == 0:
        continue
    for name, o in orders(lin).items():
        for dd in ("left", "ideal"):
            c, st, la = walk_cost(o, src[e], dd)
            k = (name, dd)
            tot[k] = tuple(np.add(tot.get(k, (0, 0, 0)), (c, st, la)))
print(f"{a.envs} envs; cost = busiest-bank lane count summed over the two lane groups, per wave-step (2 = conflict-free)")
for (name, dd), (c, st, la) in tot.items():
    print(f"  {name:32s} dedupe {dd:5s}: wave-steps {st:7d}  mean cost {c / st:5.2f}  atomics per wave-step {la / st:5.1f}")
