"""Per-workgroup phase stamps of k_hit_list / k_ray_list (a -DPHASE_TIMING build of csrc/voxel.hip) for the SAME voxel-update call
(a) inside bench.py's rollout, i.e. behind the policy evaluation of the previous env step, and (b) repeated back to back.
Why: k_ray_list takes 31-33 us in (a) and 20 us in a back-to-back loop although the rays are nearly the same (profiles/r05_notes.md 1g).

    tools/build_voxel_variant.sh gennbv_amd/libgennbv_hip_phase.so -DPHASE_TIMING
    GENNBV_HIP_LIB=$PWD/gennbv_amd/libgennbv_hip_phase.so python tools/ray_phase_rollout.py"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

n, g = 256, 64
ns = argparse.Namespace(gpus=1, steps=1, warmup=0, envs=n, grid=g, height=240, width=320, n_steps=8, batch_size=128, n_epochs=1, frames=4, backend="hip",
                        obs="compact", target_kl="off", semantic=False, no_cpu_baseline=True, no_flat_rows=True, no_state_check=True)
algo, cfg, env = bench.build_algo(ns, "cuda:0", 0, 1)
algo._setup_learn(total_timesteps=10 ** 12)
upd = env.updater
calls = []
orig = upd.update


def rec(*a, **k):
    calls.append((a, k))
    return orig(*a, **k)


upd.update = rec
nrb = ((n + 7) // 8) * 8 * 16
stamps = upd.workspace.view(torch.int32)


def summary(tag):
    torch.cuda.synchronize()
    nwg = n * max(1, (512 + n - 1) // n)
    tail = stamps[-8 * nwg:].cpu().numpy().reshape(nwg, 8)
    pa, pb, t0 = tail[:, 0] / 100.0, tail[:, 1] / 100.0, tail[:, 3] / 100.0
    print(f"[{tag}] k_hit_list over {nwg} workgroups (us): phase A mean {pa.mean():.1f} max {pa.max():.1f} | B mean {pb.mean():.1f} max {pb.max():.1f} | "
          f"start spread {t0.max() - t0.min():.1f} | end spread {(t0 + pa + pb).max() - t0.min():.1f}")
    rt = stamps[-(8 * 512 + 8 * nrb):-8 * 512].cpu().numpy().reshape(nrb, 8)[::-1]
    live, dead = rt[rt[:, 5] == 1], rt[rt[:, 5] == 2]
    if not len(live):
        print(f"[{tag}] no k_ray_list stamps (is this a -DPHASE_TIMING build?)")
        return
    t0 = live[:, 0] / 100.0
    tall = np.concatenate([live[:, 0], dead[:, 0]]) / 100.0
    base = tall.min()
    end = t0 + live[:, 3] / 100.0 - base
    print(f"[{tag}] k_ray_list: {len(live)} live / {len(dead)} dead workgroups; wave-0 walk mean {live[:, 1].mean() / 100:.1f} max {live[:, 1].max() / 100:.1f} | "
          f"item total mean {live[:, 3].mean() / 100:.1f} max {live[:, 3].max() / 100:.1f} | counts known mean {live[:, 6].mean() / 100:.2f} max {live[:, 6].max() / 100:.2f} | "
          f"pose + clear mean {live[:, 7].mean() / 100:.2f} max {live[:, 7].max() / 100:.2f}")
    print(f"[{tag}]   live starts after the first entry, deciles: {np.percentile(t0 - base, range(0, 101, 10)).round(1)}")
    print(f"[{tag}]   live ends, deciles: {np.percentile(end, range(0, 101, 10)).round(1)}")
    xcc = live[:, 4] & 15
    print(f"[{tag}]   per XCC: live workgroups {[int((xcc == x).sum()) for x in range(8)]}, last end {[round(float(end[xcc == x].max()), 1) if (xcc == x).any() else 0 for x in range(8)]}")


algo.collect_rollouts(env, None, algo.rollout_buffer, n_rollout_steps=8)
summary("inside the rollout")
a, k = calls[-1]
upd.update = orig
for _ in range(20):
    orig(*a, **k)
summary("back to back, same call")
