"""GPU: the whole C5 path on the gfx950 kernels -- ReplayFeedEnv (HIP state encoding, in-place observation rows), fused
policy head, one-launch bootstrap + add (gnbv_rollout_add), HIP GAE -- against the reference's own collect_rollouts
driving the reference's own env (fixture F11, oracle/gen_golden_rollout.py).  The GPU sampler draws from a different
random stream than torch's CPU Categorical, so the reference's sampled actions are forced; everything downstream of
the actions must then reproduce: every observation row bit for bit, env rewards bit for bit, values / log-probs /
bootstrapped rewards / advantages / returns within fp32 round-off, episode_starts exactly."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests import rollout_util as ru
from tests.test_envstep_gpu import make_env_from_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _force_actions(algo, fx):
    """policy.forward -> (the reference's action of this call, V, log pi(action)) from the HIP head's own logits."""
    pol = algo.policy
    t = int(fx["T"])
    seq = iter([fx[f"r{r}/actions_in"][s] for r in range(2) for s in range(t)])

    def forward(obs, deterministic=False):
        fused = pol._fused_head(obs)
        assert fused is not None, "the rollout must run on the fused gfx950 head"
        logits, values = fused
        a = torch.from_numpy(next(seq).astype(np.int64)).to(DEV)
        lp = pol.action_dist.proba_distribution(logits).log_prob(a.float())
        return a, values.unsqueeze(1), lp
    pol.forward = forward


@pytest.mark.parametrize("fused_add", ["1", "0"])
def test_hip_rollout_reproduces_reference_collect_rollouts(fused_add, monkeypatch):
    fx = gu.load("F11_rollout")
    env, cfg = make_env_from_fixture(fx)
    t, n = int(fx["T"]), int(fx["n"])
    algo = ru.make_algo(env, DEV, "hip", t)
    algo.fused_add = fused_add == "1"
    algo._setup_learn(total_timesteps=10 ** 9)  # env.reset() into buffer row 0 (frame 0), episode starts = 1
    reset_rows = ru.unpack_rows(fx["reset_state"], fx["reset_grid"], fx["reset_rgb"])
    assert np.array_equal(algo._last_obs.cpu().numpy(), reset_rows)
    env.episode_length_buf = torch.from_numpy(fx["init_episode_length"].astype(np.int64)).to(DEV)
    _force_actions(algo, fx)
    for r in range(2):
        assert algo.collect_rollouts(env, None, algo.rollout_buffer, n_rollout_steps=t)
        buf = algo.rollout_buffer
        rows = ru.unpack_rows(fx[f"r{r}/obs_state"], fx[f"r{r}/obs_grid"], fx[f"r{r}/obs_rgb"])
        assert np.array_equal(buf.observations[:t].cpu().numpy(), rows), "observation rows (HIP env, written in place)"
        last = ru.unpack_rows(fx[f"r{r}/last_state"], fx[f"r{r}/last_grid"], fx[f"r{r}/last_rgb"])
        assert np.array_equal(buf.observations[t].cpu().numpy(), last)
        ru.check_rollout(fx, r, buf, algo, value_tol=2e-5)
    assert algo.num_timesteps == int(fx["num_timesteps"])

