"""GPU: the BASELINE configurations at their FULL sizes on one MI355X.

* configs[1] (256 envs x 240x320 x 64^3): one collect_rollouts() of the benchmarked algorithm object (compact int8 rows, fused
  rollout) -- sampled envs replayed through the CPU oracle from an episode boundary (tests/state_check.py: every stored
  observation row, rewards, dones, the final probability / scanned grids, bit for bit), then train() on that buffer against
  the fp64 CPU loop of the same class (the statement proven equal to the reference's train() on F9).
* configs[4]'s per-GPU shard (512 envs x 240x320 x 128^3): the voxel update's size-independent properties plus a 3-env
  bit-exact oracle spot check, on the coded path the env uses.

Reference: gennbv/env/env_train_gennbv.py:277-326 (update_occ_grid), :246-264 / :346-457 (step, rewards, termination),
stable_baselines3/common/on_policy_algorithm_grid_obs.py:128-221 (collect_rollouts), ppo/ppo_grid_obs.py:196-275 (train)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SAMPLED = (0, 101, 255)


@pytest.fixture(scope="module")
def rec_full():
    from tests import test_ppo_g64_gpu as t64

    def desync(algo):
        # sampled envs time out at the first / second / third env step of the rollout, so that the oracle replay starts from an
        # episode boundary inside the 6-step buffer (the other envs keep _setup_learn's random phases)
        elb = algo.env.episode_length_buf
        for k, e in enumerate(SAMPLED):
            elb[e] = algo.env.max_episode_length - 1 - k
    return t64._Recorded(n_envs=256, t=6, hw=(240, 320), epochs=1, max_episode_length=100, frames=4, before_rollout=desync)


def test_config1_full_size_rollout_vs_oracle_env(rec_full):
    from tests import state_check
    algo = rec_full.algo
    assert algo.n_envs == 256 and algo.env.cfg.camera_height == 240 and algo.env.grid_size == 64
    assert algo.rollout_buffer.compact_state_dim is not None and getattr(algo.policy, "_fused_rollout", False)
    res = state_check.check_rollout(algo, SAMPLED)
    assert res["status"] == "bit-exact", res
    assert res["steps_replayed"] == 5 + 4 + 3 and res["episode_ends_seen"] == 0 and res.get("rewards_compared", 0) >= 6, res
    # the rollout as a whole: every env saw six steps, GAE ran, nothing non-finite
    buf = algo.rollout_buffer
    for k in ("rewards", "values", "log_probs", "advantages", "returns"):
        assert bool(torch.isfinite(getattr(buf, k)).all()), k
    tri = buf.grid_i8[1:]
    assert int(tri.min()) >= -1 and int(tri.max()) <= 1 and bool((tri != 0).flatten(2).any(2).all())


def test_config1_all_256_envs_across_episode_ends_vs_oracle_env():
    """EVERY env of one full-size rollout (256 x 240x320 x 64^3, compact rows, the prepared policy evaluation)
    replayed through the oracle env, ACROSS episode ends: episodes of 3 steps inside an 8-step buffer, so each env is replayed from its
    first boundary through one or two further done -> deferred zeroing -> next-update hand-overs (env_train_gennbv.py:377-436) -- the
    path that so far was only covered at fixture size.  Envs sharing a start row are one multi-env oracle (OpenMP over the envs)."""
    from tests import state_check
    from tests import test_ppo_g64_gpu as t64
    rec = t64._Recorded(n_envs=256, t=8, hw=(240, 320), epochs=1, max_episode_length=3, frames=4)
    algo = rec.algo
    res = state_check.check_rollout(algo, range(256))
    assert res["status"] == "bit-exact", res
    assert len(res["envs"]) == 256 and all("skipped" not in r for r in res["envs"]), res["envs"][:4]
    assert {r["from_step"] for r in res["envs"]} == {1, 2, 3}, "_setup_learn's random phases: first time-out at row 1, 2 or 3"
    # boundaries every 3 rows: the env starting at row 1 ends again at rows 4 and 7, at row 2: 5 (and 8 = the last row), at row 3: 6
    assert res["episode_ends_seen"] >= 256 and res["steps_replayed"] >= 256 * 5 and res.get("rewards_compared", 0) > 0, res
    del rec, algo
    torch.cuda.empty_cache()


def test_config1_full_size_train_vs_fp64_loop(rec_full):
    """One epoch over the full-size buffer (256 x 6 samples = 12 minibatches of 128) against the fp64 CPU loop."""
    from tests import test_ppo_g64_gpu as t64
    ref = rec_full.oracle(None)
    hip = t64._fresh_hip(rec_full, None, True)
    hip.train()
    s_h, s_r = hip.last_train_stats, ref.last_train_stats
    assert len(s_h) == len(s_r) == 12 and int(hip._hip["opt"].step_count.item()) == 12
    names = ("policy_gradient_loss", "value_loss", "entropy_loss", "approx_kl", "clip_fraction", "loss")
    for j, nm in enumerate(names):
        d = np.abs(s_h[:, j] - s_r[:, j]) / np.maximum(1.0, np.abs(s_r[:, j]))
        assert float(d.max()) <= 1e-4, (nm, int(d.argmax()), float(d.max()))  # north_star: PPO loss within 1e-4
    sd_h, sd_r = hip.policy.state_dict(), ref.policy.state_dict()
    for k, v in sd_r.items():
        if "running" in k:
            assert float((sd_h[k].double().cpu() - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k


def test_config5_shard_voxel_update_512_envs_128cubed():
    """BASELINE configs[4] per-GPU shard: 512 envs x 240x320 x 128^3, the coded update with int8 rows (what the env runs)."""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    n, h, w, g = 512, 240, 320, 128
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=5, device=DEV)
    f0, f1 = S.make_frames(scene, cfg, 2, seed=5, with_rgba=False)
    upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, DEV,
                               max_steps_between_resets=101)
    assert upd.coded
    upd.self_clean = False  # (the masks are inspected)
    tri8 = torch.zeros(n, g ** 3, dtype=torch.int8, device=DEV)
    c2w0, c2w1 = S.c2w_from_view(f0.view, scene.env_origins), S.c2w_from_view(f1.view, scene.env_origins)
    upd.update(f0.depth_raw, f0.seg_raw, c2w0, f0.poses.contiguous(), tri_i8_out=tri8, fp32_out=False)
    hit, path = upd.masks()
    hit, path = hit.view(n, -1), path.view(n, -1)
    code1 = upd.prob_code.clone()
    # hit voxels are occupied (code 0x80), path-only voxels carry one decrement (code 1), everything else is untouched (0)
    assert bool((code1[hit] == 0x80).all()) and bool((code1[path & ~hit] == 1).all()) and bool((code1[~path & ~hit] == 0).all())
    assert bool((path | ~hit).all()) and bool(hit.any(1).float().mean() > 0.9)
    assert bool((tri8[hit] == 1).all()) and bool((tri8[path & ~hit] == -1).all()) and bool((tri8[~path & ~hit] == 0).all())
    gt = scene.grid_gt.view(n, -1) > 0
    assert torch.equal(upd.coverage_count.long(), (hit & gt).sum(1))
    cov1 = upd.coverage_count.clone()
    # idempotence of the sets: the same frame again leaves hit voxels at 1 and decrements path-only voxels once more
    upd.update(f0.depth_raw, f0.seg_raw, c2w0, f0.poses.contiguous(), tri_i8_out=tri8, fp32_out=False)
    hit2, path2 = upd.masks()
    assert torch.equal(hit, hit2.view(n, -1)) and torch.equal(path, path2.view(n, -1))
    assert bool((upd.prob_code[path & ~hit] == 2).all()) and torch.equal(upd.coverage_count, cov1)
    del hit2, path2
    # a second frame, then a reset mask: the result of a lone first step again
    upd.update(f1.depth_raw, f1.seg_raw, c2w1, f1.poses.contiguous(), tri_i8_out=tri8, fp32_out=False)
    upd.update(f0.depth_raw, f0.seg_raw, c2w0, f0.poses.contiguous(), reset_mask=torch.ones(n, dtype=torch.uint8, device=DEV),
               tri_i8_out=tri8, fp32_out=False)
    assert torch.equal(upd.prob_code, code1) and torch.equal(upd.coverage_count, cov1)
    # three envs bit-exactly against the oracle over the sequence f0, f1 (probabilities, tri-class, scanned set, coverage)
    sel = [0, 257, 511]
    upd.update(f1.depth_raw, f1.seg_raw, c2w1, f1.poses.contiguous(), tri_i8_out=tri8, fp32_out=False)
    pr = np.zeros((3, g, g, g), np.float32); sc = np.zeros_like(pr)
    kinv = S.inverse_intrinsics(h, w).numpy()
    for fr, c2w in ((f0, c2w0), (f1, c2w1)):
        dp, sp = orc.post_process_depth(fr.depth_raw[sel].cpu().numpy(), fr.seg_raw[sel].cpu().numpy())
        tri_o, cov_o = orc.update_occ_grid(dp, sp, c2w[sel].cpu().numpy(), kinv, fr.poses[sel, :3].cpu().numpy(), scene.range_gt[sel].cpu().numpy(),
                                           scene.voxel_size[sel].cpu().numpy(), scene.grid_gt[sel].cpu().numpy(), pr, sc)
    sel_t = torch.tensor(sel, device=DEV)
    lut = upd._prob_lut
    assert lut[upd.prob_code[sel_t].long()].cpu().numpy().tobytes() == pr.reshape(3, -1).tobytes()
    assert np.array_equal(tri8[sel_t].cpu().numpy(), tri_o.reshape(3, -1).astype(np.int8))
    assert np.array_equal(upd.coverage_count[sel_t].cpu().numpy(), cov_o)
    assert int(upd.code_overflow.item()) == 0
