"""GPU: ReplayFeedEnv (HIP env-step kernels) vs the reference env goldens and the oracle env."""
import hashlib
import os

import numpy as np
import pytest
import torch

from tests import golden_util as gu
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_env_from_fixture(fx):
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    n, h, w, g = int(fx["n"]), int(fx["h"]), int(fx["w"]), int(fx["g"])
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    d, seg = gu.frames(fx)
    gt = torch.from_numpy(gu.unpack_bits(fx["grid_gt_bits"], g))
    scene = S.Scene(None, None, gt, torch.from_numpy(fx["range_gt"]), torch.from_numpy(fx["voxel_size"]),
                    torch.from_numpy(fx["num_valid_voxel_gt"]), torch.from_numpy(fx["env_origins"]))
    feed = ReplayFeed.from_views(torch.from_numpy(d), torch.from_numpy(seg), torch.from_numpy(fx["rgba"]),
                                 torch.from_numpy(fx["view"]), scene.env_origins)
    feed = ReplayFeed(feed.depth_raw.to(DEV), feed.seg_raw.to(DEV), feed.rgba.to(DEV), feed.c2w.to(DEV))
    env = ReplayFeedEnv(cfg, scene, feed, DEV, max_episode_length=int(fx["max_episode_length"]))
    return env, cfg


@pytest.mark.parametrize("views", [False, True, "three launches"])  # True: ReplayFeedEnv.flag_views -- dones / time_outs as views of the kernels' bytes
@pytest.mark.parametrize("name", ["F5_envstep_g20", "F5_envstep_c0", "F5_envstep_g64"])                # (collect_rollouts); "three launches": fused_observe off
def test_replay_env_matches_reference_env_bit_exact(name, views):
    if not os.path.exists(os.path.join(gu.GOLDEN, name + ".npz")):
        pytest.skip("fixture not generated")
    fx = gu.load(name)
    env, cfg = make_env_from_fixture(fx)
    # default: step()'s head + the pose-history and gray-frame slices as ONE launch (gnbv_env_observe, round 5); the three separate
    # launches must give the same bytes
    assert env.fused_observe
    if views == "three launches":
        env.fused_observe, views = False, False
    env.flag_views = views
    prev_done = None
    nf = int(fx["num_frames"])
    env.feed.cursor = 0
    obs = env.reset()
    assert sha(obs.cpu().numpy()) == str(fx["reset_flat_obs_sha"])
    assert env.prob_grid.cpu().numpy().tobytes() == fx["reset_prob"].tobytes()
    env.episode_length_buf.copy_(torch.from_numpy(fx["init_episode_length"].astype(np.int64)))
    lazy = []
    for s in range(int(fx["num_steps"])):
        fi = (s + 1) % nf
        env.feed.cursor = fi
        obs, rew, done, info = env.step(torch.from_numpy(fx["actions"][fi]).to(DEV))
        lazy.append(info["episode"])
        if s % 3 != 1:  # read at once: this step's values (two thirds of the steps; the others are first read after the loop)
            got = [float(info["episode"][k]) for k in fx["episode_keys"]]
            np.testing.assert_allclose(got, fx["episode_info"][s], rtol=1e-6, atol=1e-9, err_msg=f"step {s}")
        assert rew.cpu().numpy().tobytes() == fx["rewards"][s].tobytes(), f"step {s}"
        assert np.array_equal(done.cpu().numpy(), fx["dones"][s].astype(bool)), f"step {s}"
        assert done.dtype == torch.bool and info["time_outs"].dtype == torch.bool
        if prev_done is not None:  # the previous step's dones are still the previous step's (they are this step's episode_starts)
            assert np.array_equal(prev_done.cpu().numpy(), fx["dones"][s - 1].astype(bool)), f"step {s}: dones of step {s - 1} overwritten"
        prev_done = done
        assert np.array_equal(info["time_outs"].cpu().numpy(), fx["time_outs"][s]), f"step {s}"
        # the fixture read reward_ratio_buf[-1] after reset_idx had zeroed the reset envs' entries
        assert env.prev_ratio.cpu().numpy().tobytes() == fx["coverage"][s].tobytes()
        assert sha(obs.cpu().numpy()) == str(fx["flat_obs_sha"][s]), f"step {s} flat obs"
        assert sha(env.prob_grid.cpu().numpy()) == str(fx["prob_sha"][s])
        assert sha(env.scanned_gt_grid.cpu().numpy()) == str(fx["scan_sha"][s])
    # extras["episode"] read AFTER the loop.  The reference creates a new dict only on steps where some env resets
    # (reset_idx :424) and mutates it in place in between (base:638-639), so the entry of step s shows the values of the
    # last step that shared its dict.  Every dict accessor must fill the entry (BestCKPTCallback asserts `key in buf[0]`
    # on a never-read one).
    gen = fx["episode_dict_generation"]
    for s, e in enumerate(lazy):
        last = max(i for i in range(len(gen)) if gen[i] == gen[s])
        if s % 2:
            assert "episode_reward" in e and len(e) == len(fx["episode_keys"]) and bool(e)
        got = [float(e[k]) for k in fx["episode_keys"]] if s % 4 else [float(dict(e.items())[k]) for k in fx["episode_keys"]]
        np.testing.assert_allclose(got, fx["episode_info"][last], rtol=1e-6, atol=1e-9, err_msg=f"step {s} (alias of {last})")


def test_replay_env_writes_into_caller_rows_and_tracks_episodes():
    """obs_out = rows of a larger buffer (rollout-buffer hand-off); ring-buffer episode stats
    equal a host recomputation in env order (update_extra_episode_info)."""
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    n, h, w, g = 6, 48, 64, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=2)
    feed = ReplayFeed.synthetic(scene, cfg, 3, seed=2)
    feed = ReplayFeed(feed.depth_raw.to(DEV), feed.seg_raw.to(DEV), feed.rgba.to(DEV), feed.c2w.to(DEV))
    env = ReplayFeedEnv(cfg, scene, feed, DEV, max_episode_length=4)
    buf = torch.full((9, n, cfg.obs_dim), float("nan"), device=DEV)
    env.reset(obs_out=buf[0])
    env.episode_length_buf.copy_(torch.tensor([0, 1, 2, 3, 0, 2], device=DEV))
    g_ = torch.Generator().manual_seed(0)
    rew_host, len_host = [], []
    # reset() itself runs get_step_return -> update_extra_episode_info (reference :229-244,:346-375)
    cur_r, cur_l = env.rew_buf.cpu().numpy().copy(), np.ones(n, np.float32)
    for t in range(8):
        a = S.sample_actions(n, cfg, g_).to(DEV)
        obs, rew, done, info = env.step(a, obs_out=buf[t + 1])
        assert obs.data_ptr() == buf[t + 1].data_ptr()
        r, d = rew.cpu().numpy(), done.cpu().numpy()
        cur_r += r; cur_l += 1
        for e in range(n):
            if d[e]:
                rew_host.append(cur_r[e]); len_host.append(cur_l[e]); cur_r[e] = 0; cur_l[e] = 0
    assert not bool(torch.isnan(buf).any())
    ei = env.episode_info()
    assert len(rew_host) > 4
    np.testing.assert_allclose(ei["episode_reward"], np.mean(rew_host[-100:]), rtol=1e-6)
    np.testing.assert_allclose(ei["episode_length"], np.mean(len_host[-100:]), rtol=1e-6)
    assert int(env.ring_state.item()) == len(rew_host)


@pytest.mark.parametrize("depth_dtype", ["f32", "f16"])
def test_env_over_a_recorded_feed_file(tmp_path, depth_dtype):
    """an env fed from the on-disk container (env/feed_file.py) steps exactly like one fed from the
    same frames in memory (f32: identical tensors; f16: the in-memory feed is rounded the same way)."""
    from gennbv_amd.env import feed_file as FF
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    n, h, w, g = 5, 48, 64, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=4)
    frames = S.make_frames(scene, cfg, 5, seed=4)
    path = str(tmp_path / "rec.gnbv")
    FF.record(path, scene, frames, S.inverse_intrinsics(h, w, cfg.horizontal_fov), depth_dtype=depth_dtype)
    depth = torch.stack([f.depth_raw for f in frames])
    if depth_dtype == "f16":
        depth = depth.half().float()
    mem = ReplayFeed.from_views(depth.to(DEV), torch.stack([f.seg_raw for f in frames]).to(DEV),
                                torch.stack([f.rgba for f in frames]).to(DEV), torch.stack([f.view for f in frames]).to(DEV),
                                scene.env_origins.to(DEV))
    a_env = ReplayFeedEnv(cfg, scene, mem, DEV, max_episode_length=4)
    b_env = ReplayFeedEnv.from_file(TaskConfig(), path, DEV, max_episode_length=4)
    assert b_env.grid_size == g and b_env.num_envs == n
    oa, ob = a_env.reset(), b_env.reset()
    assert torch.equal(oa, ob)
    gen = torch.Generator().manual_seed(1)
    for _ in range(7):
        act = S.sample_actions(n, cfg, gen).to(DEV)
        ra, rb = a_env.step(act), b_env.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2])


def test_evaluate_policy_loop_on_the_replay_env():
    """evaluation.py:136-355 for tensor envs: one episode per env, AUC of the per-step reward curve, accuracies."""
    from gennbv_amd.env.replay_feed import ReplayFeed, ReplayFeedEnv
    from gennbv_amd.eval import evaluate_policy_grid_obs, mean_auc
    n, h, w, g, L = 5, 48, 64, 16, 4
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=3)
    feed = ReplayFeed.synthetic(scene, cfg, 3, seed=3)
    feed = ReplayFeed(feed.depth_raw.to(DEV), feed.seg_raw.to(DEV), feed.rgba.to(DEV), feed.c2w.to(DEV))
    env = ReplayFeedEnv(cfg, scene, feed, DEV, max_episode_length=L)
    gen = torch.Generator().manual_seed(0)
    seen = []

    class _Model:
        @staticmethod
        def policy(obs, deterministic=True):
            a = S.sample_actions(n, cfg, gen).to(DEV)
            return a, None, None

    orig_step = env.step

    def step(a):
        out = orig_step(a)
        seen.append((out[1].cpu().clone(), out[2].cpu().clone()))
        return out
    env.step = step
    rews, lens, auc, acc = evaluate_policy_grid_obs(_Model, env, n_eval_episodes=n, max_length=L, accuracy_fn=lambda i: 10.0 + i)
    assert len(rews) == len(lens) == len(acc) == n and auc.shape == (n,)
    assert sorted(acc) == [10.0 + i for i in range(n)]
    # host recomputation of the AUC curve from the recorded (reward, done) stream
    curve, flag = torch.zeros(n, L), torch.zeros(n)
    tot = torch.zeros(n)
    for t, (r, d) in enumerate(seen[:L]):
        for e in range(n):
            if flag[e]:
                curve[e, t] = curve[e, t - 1]
            elif not d[e]:
                curve[e, t] = r[e]
        tot += r * (flag == 0)
        flag += d.float()
    assert torch.allclose(auc, mean_auc(curve))
    assert all(1 <= x <= L for x in lens)


def test_eval_env_accuracy_metric_and_five_tuple():
    """Env_Eval_GenNBV protocol (env_eval_gennbv.py:104-111,:150-164,:253-263): 5-tuples, per-episode point
    accumulation, accuracy = 100 x chamfer(unique(round(points, 2)), GT cloud) for the first finished episode of an env,
    recomputed here from the same frames with the float64 oracle."""
    from oracle import oracle as orc
    from gennbv_amd import utils as U
    from gennbv_amd.env.replay_feed import ReplayFeed
    from gennbv_amd.env.replay_feed_eval import ReplayFeedEvalEnv, gt_cloud_from_grid
    from gennbv_amd.eval import evaluate_policy_grid_obs
    n, h, w, g, L = 3, 48, 64, 16, 3
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=6)
    feed = ReplayFeed.synthetic(scene, cfg, 4, seed=6)
    feed = ReplayFeed(feed.depth_raw.to(DEV), feed.seg_raw.to(DEV), feed.rgba.to(DEV), feed.c2w.to(DEV))
    env = ReplayFeedEvalEnv(cfg, scene, feed, DEV, max_episode_length=L)
    out = env.reset()  # frame 0
    assert len(out) == 5 and out[0].shape == (n, cfg.obs_dim) and out[4] == {}
    gen = torch.Generator().manual_seed(0)
    done_step = {}
    for t in range(1, 8):  # step t consumes frame t % 4
        o = env.step(S.sample_actions(n, cfg, gen).to(DEV))
        assert len(o) == 5
        for e in torch.nonzero(o[2]).flatten().tolist():
            done_step.setdefault(e, t)
        if len(done_step) == n:
            break
    assert set(env.ratios_accuracy) == {str(e) for e in range(n)}
    kinv = S.inverse_intrinsics(h, w, cfg.horizontal_fov)
    pc = gt_cloud_from_grid(scene.grid_gt, scene.range_gt, scene.voxel_size)
    for e in range(n):
        pts = []
        for f in range(0, done_step[e] + 1):
            d, s_ = U.post_process_depth(feed.depth_raw[f % 4], feed.seg_raw[f % 4])
            pts.append(U.back_projection_fg(d, s_, feed.c2w[f % 4], kinv)[e])
        cloud = torch.cat(pts, 0).cpu().numpy()
        xr = np.unique(np.round(cloud.astype(np.float32) * np.float32(100.0)) / np.float32(100.0), axis=0)
        ref = 100.0 * orc.chamfer_distance_ref(xr, pc[e].numpy())
        assert abs(env.ratios_accuracy[str(e)] - ref) <= 1e-4 * ref + 1e-6, (e, env.ratios_accuracy[str(e)], ref)

    # the evaluation loop consumes the 5-tuples
    env2 = ReplayFeedEvalEnv(cfg, scene, ReplayFeed(feed.depth_raw, feed.seg_raw, feed.rgba, feed.c2w), DEV, max_episode_length=L)

    class _Model:
        @staticmethod
        def policy(obs, deterministic=True):
            return S.sample_actions(n, cfg, gen).to(DEV), None, None

    rews, lens, auc, acc = evaluate_policy_grid_obs(_Model, env2, n_eval_episodes=n, max_length=L)
    assert len(acc) == n and all(a > 0 for a in acc) and auc.shape == (n,)


def test_best_checkpoint_callback_over_real_env_infos(tmp_path):
    """gennbv/callback.py:25-70 driven by the env's own (lazy) extras["episode"] entries: the value equals what the
    reference's callback computed over the reference env's buffer (F5 c0: 12 episode ends in 30 steps, 10 dict generations)."""
    from collections import deque
    from gennbv_amd.callback import BestCKPTCallback
    fx = gu.load("F5_envstep_c0")
    env, cfg = make_env_from_fixture(fx)
    nf = int(fx["num_frames"])
    env.feed.cursor = 0
    env.reset()
    env.episode_length_buf.copy_(torch.from_numpy(fx["init_episode_length"].astype(np.int64)))
    saved = []

    class _Model:
        num_timesteps = 0
        ep_info_buffer = deque(maxlen=100)
        logger = type("L", (), {})()

        def save(self, path):
            saved.append(os.path.basename(path))
    m = _Model()
    cb = BestCKPTCallback(save_freq=10 ** 9, save_path=str(tmp_path), name_prefix="t", key_list=["episode_reward"])
    cb.init_callback(m)
    steps = int(fx["num_steps"])
    for s in range(steps):
        env.feed.cursor = (s + 1) % nf
        _, _, _, info = env.step(torch.from_numpy(fx["actions"][(s + 1) % nf]).to(DEV))
        m.ep_info_buffer.append(info["episode"])  # never read before the callback looks at it
    cb.on_rollout_end()
    assert saved == ["t_0_steps", "t_best_episode_reward"]  # (n_calls = 0: the periodic branch fires too, as in the reference)
    # the reference's own calculate_value over its own (aliased) ep_info_buffer, recorded by oracle/gen_golden.py
    assert abs(cb.key_highest_value["episode_reward"] - float(fx["best_ckpt_value_episode_reward"])) < 1e-5
