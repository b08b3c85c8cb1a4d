"""Helpers of the collect_rollouts parity tests (fixture F11, oracle/gen_golden_rollout.py)."""
import numpy as np
import torch

from tests import golden_util as gu

G = 20


def unpack_rows(state, grid, rgb):
    return np.concatenate([state.astype(np.float32), grid.astype(np.float32), rgb.astype(np.float32)], axis=-1)


class RecordedEnv:
    """Replays the observations / rewards / dones / time_outs the REFERENCE env produced (the fixture), asserting that
    the algorithm hands it exactly the actions the reference's policy sampled (pins the RNG consumption order)."""

    def __init__(self, fx, device="cpu", check_actions=True):
        from tests import policy_util as pu
        self.fx, self.device, self.check_actions = fx, device, check_actions
        self.num_envs = int(fx["n"])
        self.max_episode_length = int(fx["max_episode_length"])
        self.observation_space, self.action_space = pu.spaces(G)
        self.episode_length_buf = torch.zeros(self.num_envs, dtype=torch.long, device=device)
        self.k = 0  # global step counter over both rollouts
        t = int(fx["T"])
        # observation returned by step k of rollout r = the row the NEXT transition starts from
        self.next_obs, self.rew, self.done, self.to, self.act = [], [], [], [], []
        for r in range(2):
            rows = unpack_rows(fx[f"r{r}/obs_state"], fx[f"r{r}/obs_grid"], fx[f"r{r}/obs_rgb"])
            last = unpack_rows(fx[f"r{r}/last_state"], fx[f"r{r}/last_grid"], fx[f"r{r}/last_rgb"])
            for s in range(t):
                self.next_obs.append(rows[s + 1] if s + 1 < t else last)
                self.rew.append(fx[f"r{r}/env_rewards"][s]); self.done.append(fx[f"r{r}/dones"][s])
                self.to.append(fx[f"r{r}/time_outs"][s]); self.act.append(fx[f"r{r}/actions_in"][s])

    def seed(self, s):
        pass

    def reset(self):
        fx = self.fx
        return torch.from_numpy(unpack_rows(fx["reset_state"], fx["reset_grid"], fx["reset_rgb"])).to(self.device)

    def step(self, actions):
        k = self.k
        self.k += 1
        if self.check_actions:
            assert np.array_equal(actions.cpu().numpy().astype(np.int64), self.act[k]), f"step {k}: sampled actions differ from the reference's"
        dev = self.device
        info = {"time_outs": torch.from_numpy(self.to[k].astype(bool)).to(dev), "episode": {"episode_reward": 0.0, "episode_length": 0.0}}
        return (torch.from_numpy(self.next_obs[k]).to(dev), torch.from_numpy(self.rew[k].copy()).to(dev),
                torch.from_numpy(self.done[k].astype(bool)).to(dev), info)


def make_algo(env, device, backend, n_steps):
    from tests import policy_util as pu
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    algo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, env, learning_rate=1e-4, n_steps=n_steps, batch_size=8, n_epochs=1, gamma=0.99,
                        gae_lambda=0.95, policy_kwargs=pu.policy_kwargs(G, backend=backend), device=device, seed=None)
    shapes = {k: tuple(v.shape) for k, v in algo.policy.state_dict().items()}
    algo.policy.load_state_dict({k: torch.from_numpy(v).to(device) for k, v in gu.det_state_dict(shapes).items()})
    return algo


def check_rollout(fx, r, buf, algo, value_tol, exact_env_rewards=True):
    """Every rollout-buffer array of rollout r against the reference's."""
    t, n = int(fx["T"]), int(fx["n"])
    f = lambda x: x.detach().cpu().numpy().reshape(t, n, -1).squeeze(-1) if x.dim() == 3 and x.shape[-1] == 1 else x.detach().cpu().numpy()  # noqa: E731
    assert np.array_equal(f(buf.actions), fx[f"r{r}/actions"]), "actions"
    assert np.array_equal(f(buf.episode_starts).astype(np.uint8), fx[f"r{r}/episode_starts"]), "episode_starts hand-over"
    np.testing.assert_allclose(f(buf.values), fx[f"r{r}/values"], rtol=0, atol=value_tol, err_msg="values")
    np.testing.assert_allclose(f(buf.log_probs), fx[f"r{r}/log_probs"], rtol=0, atol=value_tol * 4, err_msg="log_probs")
    rew, env_rew = f(buf.rewards), fx[f"r{r}/env_rewards"]
    boot = fx[f"r{r}/time_outs"].astype(bool)
    # rows without a time-out: the env's reward untouched (bit-exact); with one: + gamma * V(new_obs)[ENV 0] (:205-208)
    assert np.array_equal(rew[~boot], env_rew[~boot]) and np.array_equal(fx[f"r{r}/rewards"][~boot], env_rew[~boot])
    np.testing.assert_allclose(rew, fx[f"r{r}/rewards"], rtol=0, atol=value_tol, err_msg="bootstrapped rewards")
    assert boot.sum() > 0 and np.abs(fx[f"r{r}/rewards"][boot] - env_rew[boot]).min() > 1e-3  # the bootstrap is exercised
    np.testing.assert_allclose(f(buf.advantages), fx[f"r{r}/advantages"], rtol=0, atol=value_tol * 20, err_msg="advantages")
    np.testing.assert_allclose(f(buf.returns), fx[f"r{r}/returns"], rtol=0, atol=value_tol * 20, err_msg="returns")
    assert np.array_equal(np.asarray(algo._last_episode_starts.cpu() if torch.is_tensor(algo._last_episode_starts) else algo._last_episode_starts)
                          .astype(np.uint8), fx[f"r{r}/last_episode_starts"])
