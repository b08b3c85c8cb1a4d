"""Generate the rollout / evaluation goldens by RUNNING THE REFERENCE (this container only).

TEST INFRASTRUCTURE ONLY.   python oracle/gen_golden_rollout.py

  F11_rollout  C5  the reference's own `collect_rollouts` (stable_baselines3/common/on_policy_algorithm_grid_obs.py:128-221)
               driving the reference's own Env_Train_GenNBV.step() on the fake simulator (the F5 set-up) with the
               reference's own policy (G = 20, deterministic weights), Categorical sampling seeded on the CPU:
               two consecutive rollouts (hand-over of `_last_obs` / `_last_episode_starts`), time-outs inside
               (bootstrap :205-208 -- NOTE the `[0]`: every env is bootstrapped with env 0's value), the final
               predict_values (:213-217), GAE.  Stored: the feed, every observation row (packed), the sampled actions,
               the env's raw rewards / dones / time_outs, and every rollout-buffer array.
  F14_eval_callback  f3  `EvalCallback_Grid_Obs` (stable_baselines3/common/callbacks.py:473-708) driven over a scripted env
  F12_eval     f3  `evaluate_policy_grid_obs` + `AUC_update` (stable_baselines3/common/evaluation.py:136-378) over a
               scripted 50-env 5-tuple env and a scripted model.predict: episode rewards / lengths / accuracies in
               the order the reference emits them and the mean-AUC vector.
"""
from __future__ import annotations

import os
import sys
import types
from collections import deque

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
import gen_golden as gg  # noqa: E402
import gen_golden_ppo as gp  # noqa: E402
from gennbv_amd.env import synthetic as S  # noqa: E402
from gennbv_amd.env.config import TaskConfig  # noqa: E402

GOLDEN = gg.GOLDEN


class _RefEnvAdapter:
    """The tensor-env protocol collect_rollouts needs (EnvWrapperGenNBVTrain, env_wrapper_gennbv_train.py:59-110)
    around the reference Env_Train_GenNBV on the fake simulator: feeds the next recorded frame, steps, flattens."""

    def __init__(self, ref, env, frames, max_episode_length):
        self.ref, self.env, self.frames = ref, env, frames
        self.num_envs = env.num_envs
        self.max_episode_length = max_episode_length
        self.cursor = 0
        self.log = {"obs": [], "rewards": [], "dones": [], "time_outs": [], "actions_in": []}

    def _feed(self):
        d, s, r, v = self.frames[self.cursor % len(self.frames)]
        gg.feed_frame(self.env, d, s, r, v)
        self.cursor += 1

    def _flat(self, obs):
        return self.ref.wrapper.flatten_observations(obs, ["state", "grid", "state_rgb"])

    def reset(self):
        self._feed()
        return self._flat(type(self.env).reset(self.env))

    def step(self, actions):
        self._feed()
        self.log["actions_in"].append(actions.clone().numpy())
        obs, rew, done, info = type(self.env).step(self.env, actions)
        flat = self._flat(obs)
        self.log["obs"].append(flat.numpy().copy())
        self.log["rewards"].append(rew.numpy().copy())  # BEFORE collect_rollouts adds the bootstrap in place
        self.log["dones"].append(done.numpy().copy())
        self.log["time_outs"].append(info["time_outs"].numpy().copy())
        return flat, rew, done, info

    @property
    def episode_length_buf(self):
        return self.env.episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, v):
        self.env.episode_length_buf = v


def pack_rows(flat, g):
    """flat obs rows [..., 600 + g^3 + 8192] -> (state f32, grid i8, rgb u8); exact (grid in {-1,0,1}, gray in 0..255)."""
    st = flat[..., :600].astype(np.float32)
    gr = flat[..., 600:600 + g ** 3]
    rg = flat[..., 600 + g ** 3:]
    assert np.array_equal(gr, gr.astype(np.int8)) and np.array_equal(rg, rg.astype(np.uint8))
    return st, gr.astype(np.int8), rg.astype(np.uint8)


def gen_rollout(ref):
    n, h, w, g, T, n_frames, max_len, seed = 4, 60, 80, 20, 12, 3, 7, 21
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=seed)
    frames = S.make_frames(scene, cfg, n_frames, seed=seed)
    fr, depth_q, segs, rgbas, views = [], [], [], [], []
    for f in frames:
        q, d = gg.quantize_depth(f.depth_raw)
        depth_q.append(q); segs.append(f.seg_raw.numpy().astype(np.uint8)); rgbas.append(f.rgba.numpy()); views.append(f.view.numpy())
        fr.append((d, torch.from_numpy(segs[-1]).float(), torch.from_numpy(rgbas[-1]), torch.from_numpy(views[-1])))
    renv = gg.make_ref_env(ref, cfg, scene, max_len)
    gg.patch_bresenham(ref)
    env = _RefEnvAdapter(ref, renv, fr, max_len)
    pol, obs_space, act_space = gp.make_policy(ref, seed=0)
    gamma, lam = 0.99, 0.95

    Algo = ref.on_policy.OnPolicyAlgorithm_Grid_Obs if hasattr(ref.on_policy, "OnPolicyAlgorithm_Grid_Obs") else None
    if Algo is None:
        Algo = next(v for k, v in vars(ref.on_policy).items() if isinstance(v, type) and "collect_rollouts" in vars(v))
    algo = object.__new__(Algo)
    algo.policy, algo.use_sde, algo.sde_sample_freq, algo.is_isaac_gym_env = pol, False, -1, True
    algo.action_space, algo.device, algo.gamma, algo.num_timesteps = act_space, torch.device("cpu"), gamma, 0
    algo.ep_info_buffer = deque(maxlen=100)
    cb = types.SimpleNamespace(on_rollout_start=lambda: None, update_locals=lambda l: None, on_step=lambda: True, on_rollout_end=lambda: None)
    np.random.seed(77)
    buf = ref.buffers.TensorRolloutBuffer_Grid_Obs(T, obs_space, act_space, device="cpu", gamma=gamma, gae_lambda=lam, n_envs=n)
    # _setup_learn (base_class_grid_obs.py:451-475): reset, all-ones episode starts, staggered episode lengths (fixed here)
    algo._last_obs = env.reset()
    reset_obs = algo._last_obs.numpy().copy()
    algo._last_episode_starts = np.ones((n,), dtype=bool)
    init_len = torch.tensor([0, 2, 4, 5], dtype=torch.long)
    env.episode_length_buf = init_len.clone()
    torch.manual_seed(5)
    out = dict(n=n, h=h, w=w, g=g, T=T, num_frames=n_frames, max_episode_length=max_len, seed=seed, gamma=gamma, gae_lambda=lam,
               torch_seed=5, depth_q=np.stack(depth_q), seg=np.stack(segs), rgba=np.stack(rgbas), view=np.stack(views),
               special_idx=np.zeros((0,), np.int64), special_val=np.zeros((0,), np.float32),
               env_origins=scene.env_origins.numpy(), range_gt=scene.range_gt.numpy(), voxel_size=scene.voxel_size.numpy(),
               grid_gt_bits=gg.pack_bits(scene.grid_gt.numpy()), num_valid_voxel_gt=scene.num_valid_voxel_gt.numpy(),
               init_episode_length=init_len.numpy())
    st, gr, rg = pack_rows(reset_obs, g)
    out.update(reset_state=st, reset_grid=gr, reset_rgb=rg)
    for r in range(2):
        k0 = len(env.log["obs"])
        ok = Algo.collect_rollouts(algo, env, cb, buf, n_rollout_steps=T)
        assert ok and buf.full
        obs_rows = buf.observations.numpy()  # [T, N, D]: the observation each transition was taken from
        st, gr, rg = pack_rows(obs_rows, g)
        new_obs = np.stack(env.log["obs"][k0:])  # what env.step returned (row t+1's content)
        st2, gr2, rg2 = pack_rows(new_obs[-1], g)
        out.update({f"r{r}/obs_state": st, f"r{r}/obs_grid": gr, f"r{r}/obs_rgb": rg,
                    f"r{r}/last_state": st2, f"r{r}/last_grid": gr2, f"r{r}/last_rgb": rg2,
                    f"r{r}/actions": buf.actions.numpy().copy(), f"r{r}/rewards": buf.rewards.numpy().reshape(T, n).copy(),
                    f"r{r}/episode_starts": buf.episode_starts.numpy().reshape(T, n).astype(np.uint8),
                    f"r{r}/values": buf.values.numpy().reshape(T, n).copy(), f"r{r}/log_probs": buf.log_probs.numpy().reshape(T, n).copy(),
                    f"r{r}/advantages": buf.advantages.numpy().reshape(T, n).copy(), f"r{r}/returns": buf.returns.numpy().reshape(T, n).copy(),
                    f"r{r}/env_rewards": np.stack(env.log["rewards"][k0:]), f"r{r}/dones": np.stack(env.log["dones"][k0:]).astype(np.uint8),
                    f"r{r}/time_outs": np.stack(env.log["time_outs"][k0:]).astype(np.uint8),
                    f"r{r}/actions_in": np.stack(env.log["actions_in"][k0:]),
                    f"r{r}/last_episode_starts": np.asarray(algo._last_episode_starts).astype(np.uint8)})
        assert np.array_equal(buf.actions.numpy(), np.stack(env.log["actions_in"][k0:]).astype(np.float32))
        print(f"rollout {r}: time_outs {int(np.stack(env.log['time_outs'][k0:]).sum())}, dones {int(np.stack(env.log['dones'][k0:]).sum())}, "
              f"bootstrapped entries {int((np.abs(buf.rewards.numpy().reshape(T, n) - np.stack(env.log['rewards'][k0:])) > 0).sum())}")
    out["num_timesteps"] = algo.num_timesteps
    np.savez_compressed(os.path.join(GOLDEN, "F11_rollout.npz"), **out)
    print("F11_rollout saved,", os.path.getsize(os.path.join(GOLDEN, "F11_rollout.npz")) // 1024, "KiB")


def gen_eval(ref):
    """Scripted 5-tuple env (50 envs -- the reference hard-codes n_envs = 50, max_length = 30, :201-202)."""
    n_envs, max_len, steps = 50, 30, 30
    rs = np.random.RandomState(9)
    rewards = rs.rand(steps, n_envs).astype(np.float32)
    # every env finishes exactly once somewhere in 3..30 (an env may also emit later dones that must be ignored)
    first_done = rs.randint(3, steps + 1, size=n_envs)
    first_done[:5] = steps  # some end on the very last step
    dones = np.zeros((steps, n_envs), np.int64)
    for i in range(n_envs):
        dones[first_done[i] - 1, i] = 1
        if first_done[i] + 4 <= steps:
            dones[first_done[i] + 3, i] = 1  # a second episode end of an env that is already counted
    acc = rs.rand(steps, n_envs).astype(np.float32)
    calls = {"t": 0, "obs": []}

    class Env:
        num_envs = n_envs

        def env_is_wrapped(self, cls):
            return [False]

        def reset(self):
            calls["t"] = 0
            return torch.zeros(n_envs, 4), torch.zeros(n_envs), torch.zeros(n_envs), {}, {}

        def step(self, actions):
            t = calls["t"]
            calls["t"] += 1
            a = {str(i): float(acc[t, i]) for i in range(n_envs)}
            return (torch.full((n_envs, 4), float(t + 1)), torch.from_numpy(rewards[t]), torch.from_numpy(dones[t]), {"episode": {}}, a)

    class Model:
        def predict(self, observations, state=None, deterministic=True):
            calls["obs"].append(float(observations[0, 0]))
            return torch.zeros(n_envs, 6, dtype=torch.long), None

    ev = ref.evaluation
    ev.is_vecenv_wrapped = lambda env, cls: False
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ep_r, ep_l, mean_auc, ep_acc = ev.evaluate_policy_grid_obs(Model(), Env(), n_eval_episodes=n_envs, deterministic=True)
    out = dict(n_envs=n_envs, max_length=max_len, rewards=rewards, dones=dones, accuracies=acc,
               episode_rewards=np.array([float(x) for x in ep_r], np.float64), episode_lengths=np.array([int(x) for x in ep_l], np.int64),
               mean_auc=mean_auc.numpy().astype(np.float32), episode_accuracies=np.array([float(x) for x in ep_acc], np.float64),
               steps_run=calls["t"])
    # AUC_update alone on a second script where some envs never finish inside max_length
    auc = torch.zeros(7, 6)
    flag = torch.zeros(7)
    rs = np.random.RandomState(3)
    r2 = rs.rand(6, 7).astype(np.float32)
    d2 = (rs.rand(6, 7) < 0.25).astype(np.int64)
    snaps = []
    for t in range(6):
        auc = ev.AUC_update(auc.clone(), torch.from_numpy(r2[t]).clone(), t + 1, torch.from_numpy(d2[t]), flag)
        flag = flag + torch.from_numpy(d2[t]).float()
        snaps.append(auc.numpy().copy())
    out.update(auc2_rewards=r2, auc2_dones=d2, auc2_snapshots=np.stack(snaps))
    np.savez_compressed(os.path.join(GOLDEN, "F12_eval.npz"), **out)
    print("F12_eval saved; episodes", len(ep_r), "steps", calls["t"])


def evalcb_script():
    """The scripted eval env of F14 (shared by the generator and tests/test_eval_cpu.py through the fixture's arrays)."""
    n_envs, steps = 50, 30
    rs = np.random.RandomState(21)
    rewards = rs.rand(steps, n_envs).astype(np.float32)
    first_done = rs.randint(2, steps + 1, size=n_envs)
    dones = np.zeros((steps, n_envs), np.int64)
    for i in range(n_envs):
        dones[first_done[i] - 1, i] = 1
    acc = rs.rand(steps, n_envs).astype(np.float32)
    # one reward scale per evaluation: new bests at evaluations 0, 2, 5 (3 repeats 2's value: not strictly better)
    scales = np.array([1.0, 0.6, 1.3, 1.3, 0.9, 2.0, 0.5], np.float32)
    success_from = 2  # infos carries "is_success" from this evaluation on
    return n_envs, steps, rewards, dones, acc, scales, success_from


def gen_evalcb(ref):
    """F14: the reference's own EvalCallback_Grid_Obs (stable_baselines3/common/callbacks.py:473-708) driven for 21
    on_step() calls at eval_freq = 3 over a scripted 50-env eval env: what it records, dumps, saves and returns."""
    import importlib
    import tempfile
    cbm = importlib.import_module("stable_baselines3.common.callbacks")
    n_envs, steps, rewards, dones, acc, scales, success_from = evalcb_script()
    st = {"t": 0, "k": -1}

    class Env:
        num_envs = n_envs

        def env_is_wrapped(self, cls):
            return [False]

        def reset(self):
            st["t"] = 0
            st["k"] += 1
            return torch.zeros(n_envs, 4), torch.zeros(n_envs), torch.zeros(n_envs), {}, {}

        def step(self, actions):
            t, k = st["t"], st["k"]
            st["t"] += 1
            infos = {"episode": {}}
            if k >= success_from:
                infos["is_success"] = float((t + k) % 3 == 0)
            return (torch.full((n_envs, 4), float(t + 1)), torch.from_numpy(rewards[t] * scales[k]), torch.from_numpy(dones[t]), infos,
                    {str(i): float(acc[t, i]) for i in range(n_envs)})

    log = {"records": [], "dumps": [], "saves": []}

    class Logger:
        def record(self, key, value, exclude=None):
            log["records"].append((key, float(value), "" if exclude is None else str(exclude)))

        def dump(self, step=0):
            log["dumps"].append(int(step))

    class Model:
        num_timesteps = 0
        logger = Logger()

        def get_env(self):
            return Env()

        def get_vec_normalize_env(self):
            return None

        def predict(self, observations, state=None, deterministic=True):
            return torch.zeros(n_envs, 6, dtype=torch.long), None

        def save(self, path):
            log["saves"].append((os.path.basename(path), int(self.num_timesteps)))

    class Count(cbm.BaseCallback):
        def __init__(self, stop_at=None):
            super().__init__()
            self.stop_at, self.seen = stop_at, []

        def _on_step(self):
            self.seen.append(int(self.num_timesteps))
            return not (self.stop_at is not None and len(self.seen) == self.stop_at)

    ev = ref.evaluation
    ev.is_vecenv_wrapped = lambda env, cls: False
    cbm.sync_envs_normalization = lambda a, b: None
    import warnings
    with tempfile.TemporaryDirectory() as d, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        on_best, after = Count(), Count(stop_at=5)
        cb = cbm.EvalCallback_Grid_Obs(Env(), callback_on_new_best=on_best, callback_after_eval=after, n_eval_episodes=n_envs, eval_freq=3,
                                       log_path=os.path.join(d, "log"), best_model_save_path=os.path.join(d, "best"), verbose=0)
        st["k"] = -1
        model = Model()
        cb.init_callback(model)
        rets = []
        for c in range(21):
            model.num_timesteps += n_envs
            rets.append(bool(cb.on_step()))
        z = np.load(os.path.join(d, "log", "evaluations.npz"))
        npz = {k: np.asarray(z[k]) for k in z.files}
        assert on_best.parent is cb and after.parent is cb
    keys = sorted({r[0] for r in log["records"]})
    out = dict(n_envs=n_envs, steps=steps, rewards=rewards, dones=dones, accuracies=acc, scales=scales, success_from=success_from,
               eval_freq=3, n_calls=21, returns=np.array(rets), record_keys=np.array([r[0] for r in log["records"]]),
               record_values=np.array([r[1] for r in log["records"]], np.float64), record_exclude=np.array([r[2] for r in log["records"]]),
               dumps=np.array(log["dumps"], np.int64), save_names=np.array([s[0] for s in log["saves"]]),
               save_timesteps=np.array([s[1] for s in log["saves"]], np.int64), on_best_seen=np.array(on_best.seen, np.int64),
               after_seen=np.array(after.seen, np.int64), best_mean_reward=float(cb.best_mean_reward), last_mean_reward=float(cb.last_mean_reward),
               **{"npz_" + k: v for k, v in npz.items()})
    np.savez_compressed(os.path.join(GOLDEN, "F14_eval_callback.npz"), **out)
    print("F14_eval_callback saved; keys", keys, "saves", log["saves"], "returns", rets, "npz", {k: v.shape for k, v in npz.items()})


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    from build_ref import build as build_ref
    build_ref()
    ref = ref_harness.import_reference()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["rollout", "eval", "evalcb"]
    if "rollout" in which:
        gen_rollout(ref)
    if "eval" in which:
        gen_eval(ref)
    if "evalcb" in which:
        gen_evalcb(ref)
