"""CPU: the oracle (oracle/oracle.c) against the golden vectors produced by the reference."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests import golden_util as gu
from gennbv_amd.env import synthetic as S


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_bresenham_matches_reference_kernel_goldens():
    fx = gu.load("F4_bresenham")
    keys = sorted(k[:-4] for k in fx.files if k.endswith("_src"))
    assert len(keys) == 15
    for k in keys:
        g = int(k.split("_")[0][1:])
        traj, lens = orc.bresenham3d(fx[k + "_src"], fx[k + "_tgt"], g)
        assert np.array_equal(lens, fx[k + "_len"]), k
        assert np.array_equal(traj, fx[k + "_traj"].astype(np.int32)), k


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built")
def test_bresenham_restatement_equals_compiled_reference_random():
    rs = np.random.RandomState(0)
    for g in (8, 16, 20, 33, 64):
        for _ in range(6):
            src = rs.randint(-2 * g, 3 * g, size=3).astype(np.int32)
            tgt = rs.randint(0, g, size=(200, 3)).astype(np.int32)
            a = orc.bresenham3d(src, tgt, g)
            b = orc.ref_bresenham3d(src, tgt, g)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


def test_postprocess_backprojection_voxelidx_bit_exact():
    fx = gu.load("F2_backproj")
    n, h, w, g = int(fx["n"]), int(fx["h"]), int(fx["w"]), int(fx["g"])
    d, seg = gu.frames(fx)
    dp, sp = orc.post_process_depth(d[0], seg[0])
    # A1: bit-exact incl. NaN / +-inf / < -50 specials
    assert dp.tobytes() == fx["depth_processed"].tobytes()
    assert sp.tobytes() == fx["seg_processed"].tobytes()
    # host plumbing: c2w through torch.linalg.inv exactly as env_train_gennbv.py:512-514
    c2w = S.c2w_from_view(torch.from_numpy(fx["view"]), torch.from_numpy(fx["env_origins"])).numpy()
    assert c2w.tobytes() == fx["c2w"].tobytes()
    world, fg = orc.back_projection(dp, sp, c2w, fx["inv_intri"])
    idx = orc.points_to_idx(world, fg, fx["range_gt"], fx["voxel_size"], g)
    for e in range(n):
        ref_pts = fx[f"world_{e}"]
        mine = world[e][fg[e]]
        # A2: every foreground world point bit-exact (canonical k-ordered fma chain)
        assert mine.shape == ref_pts.shape
        assert mine.tobytes() == ref_pts.tobytes(), f"env {e}: {(mine != ref_pts).sum()} coords differ"
        # A3: unique clamped voxel index set (torch.unique(dim=0) sorts rows lexicographically)
        kept = idx[e][(idx[e][:, 0] >= 0)]
        uniq = np.unique(kept.astype(np.int64), axis=0)
        assert np.array_equal(uniq, fx[f"uidx_{e}"].reshape(-1, 3))
    # A4 incl. out-of-grid poses (no clamp)
    assert np.array_equal(orc.pose_to_idx(fx["poses"][:, :3], fx["range_gt"], fx["voxel_size"]), fx["pose_idx"])
    assert np.array_equal(orc.pose_to_idx(fx["far_poses"], fx["range_gt"], fx["voxel_size"]), fx["far_idx"])
    assert (fx["far_idx"] >= g).any() and (fx["far_idx"] < 0).any()


def test_gae_both_conventions_bit_exact():
    fx = gu.load("F8_gae")
    adv, ret = orc.gae_sb3(fx["rewards"], fx["values"], fx["episode_starts"], fx["last_values"], fx["dones"])
    assert adv.tobytes() == fx["sb3_advantages"].tobytes()
    assert ret.tobytes() == fx["sb3_returns"].tobytes()
    rret, radv = orc.gae_rsl(fx["rewards"], fx["values"], fx["rsl_dones"], fx["last_values"])
    assert rret.tobytes() == fx["rsl_returns"].tobytes()
    # whole-buffer normalisation (rollout_storage.py:143-144) is a float reduction: tolerance
    norm = (radv - radv.mean()) / (radv.std(ddof=1) + 1e-8)
    np.testing.assert_allclose(norm, fx["rsl_advantages_normalized"], rtol=1e-5, atol=1e-6)


def _replay_env_fixture(name):
    """Drive oracle/env_oracle.OracleEnv with the recorded feed of a F5 fixture."""
    from oracle.env_oracle import OracleEnv
    from gennbv_amd.env.config import TaskConfig
    fx = gu.load(name)
    n, h, w, g = int(fx["n"]), int(fx["h"]), int(fx["w"]), int(fx["g"])
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    d, seg = gu.frames(fx)
    gt = gu.unpack_bits(fx["grid_gt_bits"], g)
    env = OracleEnv(cfg, fx["inv_intri"], fx["range_gt"], fx["voxel_size"], gt, fx["num_valid_voxel_gt"],
                    max_episode_length=int(fx["max_episode_length"]))
    org = torch.from_numpy(fx["env_origins"])
    c2w = [S.c2w_from_view(torch.from_numpy(v), org).numpy() for v in fx["view"]]
    return fx, env, d, seg, c2w


@pytest.mark.parametrize("name", ["F5_envstep_g20", "F5_envstep_c0", "F5_envstep_g64"])
def test_env_step_oracle_matches_reference_env(name):
    """A1-A9 end to end: rewards, dones, time_outs (incl. the stale-extras quirk), grids and the
    flat observation of every step equal the reference env's, bit for bit."""
    import os
    if not os.path.exists(os.path.join(gu.GOLDEN, name + ".npz")):
        pytest.skip("fixture not generated")
    fx, env, d, seg, c2w = _replay_env_fixture(name)
    nf = int(fx["num_frames"])
    obs = env.reset(d[0], seg[0], fx["rgba"][0], c2w[0])
    assert sha(obs) == str(fx["reset_flat_obs_sha"])
    assert env.prob_grid.tobytes() == fx["reset_prob"].tobytes()
    env.episode_length_buf = fx["init_episode_length"].astype(np.int64).copy()
    keep = set(int(k) for k in fx["keep_steps"])
    for s in range(int(fx["num_steps"])):
        fi = (s + 1) % nf
        # grids as the observation sees them (before reset_idx zeroes the reset envs)
        obs, rew, done, info = env.step(fx["actions"][fi], d[fi], seg[fi], fx["rgba"][fi], c2w[fi])
        assert rew.tobytes() == fx["rewards"][s].tobytes(), f"step {s} rewards {rew} vs {fx['rewards'][s]}"
        assert np.array_equal(done, fx["dones"][s].astype(bool)), f"step {s}"
        assert np.array_equal(info["time_outs"], fx["time_outs"][s]), f"step {s} time_outs"
        # the fixture read reward_ratio_buf[-1] after reset_idx had zeroed the reset envs' entries
        assert np.where(done, np.float32(0), info["coverage"]).tobytes() == fx["coverage"][s].tobytes()
        assert sha(obs) == str(fx["flat_obs_sha"][s]), f"step {s} flat obs"
        assert sha(env.prob_grid) == str(fx["prob_sha"][s]) and sha(env.scanned_gt_grid) == str(fx["scan_sha"][s])
        if s in keep:
            g3 = env.g ** 3
            assert np.array_equal(obs[:, 600:600 + g3].astype(np.int8).reshape(fx[f"tri_{s}"].shape), fx[f"tri_{s}"])
            assert obs[:, :600].tobytes() == fx[f"pose_state_{s}"].reshape(env.n, -1).tobytes()
    assert fx["dones"].sum() > 0 or name != "F5_envstep_c0"
    # fp32 rounding of repeated -0.05: the g20 fixture runs 40 steps, >20 decrements on some voxels
    if name == "F5_envstep_g20":
        assert (env.prob_grid < -1.0).any()
