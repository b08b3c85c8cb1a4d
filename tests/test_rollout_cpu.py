"""CPU: PPO_Grid_Obs.collect_rollouts (host logic, torch reference encoder) against the reference's own
collect_rollouts (stable_baselines3/common/on_policy_algorithm_grid_obs.py:128-221) recorded in fixture F11:
same seed -> the SAME sampled actions (RNG consumption order), every buffer row, the env-0 time-out bootstrap, the
episode_starts hand-over across two rollouts, final predict_values + GAE."""
import numpy as np
import torch

from tests import golden_util as gu
from tests import rollout_util as ru


def test_collect_rollouts_matches_reference_rows_and_rng_order(monkeypatch):
    fx = gu.load("F11_rollout")
    env = ru.RecordedEnv(fx, "cpu")
    algo = ru.make_algo(env, "cpu", "torch", int(fx["T"]))
    # the CPU path has no HIP GAE kernel: the test's numpy statement of buffers.py:706-724 (checked against F8 elsewhere)
    from oracle import oracle as orc
    from gennbv_amd import gae as gae_mod

    def gae_cpu(rewards, values, episode_starts, last_values, dones, gamma, lam, advantages=None, returns=None):
        t, n = rewards.shape[0], rewards.shape[1]
        a, r = orc.gae_sb3(rewards.reshape(t, n).numpy(), values.reshape(t, n).numpy(), episode_starts.reshape(t, n).numpy().astype(np.uint8),
                           last_values.reshape(-1).numpy(), np.asarray(dones.numpy() if torch.is_tensor(dones) else dones).astype(np.uint8), gamma, lam)
        advantages.copy_(torch.from_numpy(a).view(t, n, 1)); returns.copy_(torch.from_numpy(r).view(t, n, 1))
        return advantages, returns
    monkeypatch.setattr(gae_mod, "compute_returns_and_advantage", gae_cpu)
    algo._setup_learn(total_timesteps=10 ** 9)
    env.episode_length_buf = torch.from_numpy(fx["init_episode_length"].astype(np.int64))
    torch.manual_seed(int(fx["torch_seed"]))
    for r in range(2):
        assert algo.collect_rollouts(env, None, algo.rollout_buffer, n_rollout_steps=int(fx["T"]))
        ru.check_rollout(fx, r, algo.rollout_buffer, algo, value_tol=2e-6)
        rows = ru.unpack_rows(fx[f"r{r}/obs_state"], fx[f"r{r}/obs_grid"], fx[f"r{r}/obs_rgb"])
        assert np.array_equal(algo.rollout_buffer.observations[:int(fx["T"])].numpy(), rows)
    assert algo.num_timesteps == int(fx["num_timesteps"])


def _gae_cpu(monkeypatch):
    from oracle import oracle as orc
    from gennbv_amd import gae as gae_mod

    def gae_cpu(rewards, values, episode_starts, last_values, dones, gamma, lam, advantages=None, returns=None):
        t, n = rewards.shape[0], rewards.shape[1]
        a, r = orc.gae_sb3(rewards.reshape(t, n).numpy(), values.reshape(t, n).numpy(), episode_starts.reshape(t, n).numpy().astype(np.uint8),
                           last_values.reshape(-1).numpy(), np.asarray(dones.numpy() if torch.is_tensor(dones) else dones).astype(np.uint8), gamma, lam)
        advantages.copy_(torch.from_numpy(a).view(t, n, 1)); returns.copy_(torch.from_numpy(r).view(t, n, 1))
        return advantages, returns
    monkeypatch.setattr(gae_mod, "compute_returns_and_advantage", gae_cpu)


def test_per_env_bootstrap_is_the_documented_non_reference_option(monkeypatch):
    """`timeout_bootstrap="per_env"` (each env's own V(new_obs), SB3's intent) is NOT what the reference computes: its
    `predict_values(new_obs)[0]` (:206) bootstraps every env with env 0's value.  The default must be the reference's."""
    fx = gu.load("F11_rollout")
    env = ru.RecordedEnv(fx, "cpu")
    algo = ru.make_algo(env, "cpu", "torch", int(fx["T"]))
    assert algo.timeout_bootstrap == "reference"
    algo.timeout_bootstrap = "per_env"
    _gae_cpu(monkeypatch)
    algo._setup_learn(total_timesteps=10 ** 9)
    env.episode_length_buf = torch.from_numpy(fx["init_episode_length"].astype(np.int64))
    torch.manual_seed(int(fx["torch_seed"]))
    algo.collect_rollouts(env, None, algo.rollout_buffer, n_rollout_steps=int(fx["T"]))
    t, n = int(fx["T"]), int(fx["n"])
    rew = algo.rollout_buffer.rewards.numpy().reshape(t, n)
    boot = fx["r0/time_outs"].astype(bool)
    boot[:, 0] = False  # env 0 itself is bootstrapped with its own value in both modes
    assert np.abs(rew[boot] - fx["r0/rewards"][boot]).max() > 1e-3
