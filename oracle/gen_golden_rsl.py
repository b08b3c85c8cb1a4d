"""Generate the rsl_rl golden by RUNNING THE REFERENCE (this container only).   python oracle/gen_golden_rsl.py

TEST INFRASTRUCTURE ONLY.
  F13_rsl_ppo  C-alt  the vendored rsl_rl's own PPO (rsl_rl/algorithms/ppo.py:95-199: act / process_env_step with the
               time-out bootstrap / compute_returns / update) over its own RolloutStorage
               (rsl_rl/storage/rollout_storage.py:86-192) and its own ActorCritic (a small Gaussian MLP), seeded on the CPU:
               every storage array after the rollout, the minibatch permutation, the two mean losses, the parameters after
               6 optimizer steps.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
from tests import golden_util as gu  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    ref = ref_harness.import_reference()
    import importlib
    AC = importlib.import_module("rsl_rl.modules.actor_critic").ActorCritic
    n, t, d_obs, d_act = 8, 6, 12, 3
    cfgs = dict(num_learning_epochs=2, num_mini_batches=3, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
                learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01)
    ac = AC(d_obs, d_obs, d_act, actor_hidden_dims=[32, 32], critic_hidden_dims=[32, 32], activation="elu", init_noise_std=0.8)
    shapes = {k: tuple(v.shape) for k, v in ac.state_dict().items()}
    init = gu.det_state_dict(shapes)
    init["std"] = np.full((d_act,), 0.8, np.float32)
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in init.items()})
    ppo = ref.rsl_ppo.PPO(ac, device="cpu", **cfgs)
    ppo.init_storage(n, t, [d_obs], [None], [d_act])
    gen = torch.Generator().manual_seed(3)
    obs = torch.randn(t + 1, n, d_obs, generator=gen)
    rewards = torch.rand(t, n, generator=gen)
    dones = (torch.rand(t, n, generator=gen) < 0.25)
    time_outs = dones & (torch.rand(t, n, generator=gen) < 0.6)
    torch.manual_seed(11)
    actions = []
    for s in range(t):
        a = ppo.act(obs[s], obs[s])
        actions.append(a.numpy().copy())
        ppo.process_env_step(rewards[s].clone(), dones[s].clone(), {"time_outs": time_outs[s].clone()})
    ppo.compute_returns(obs[t])
    st = ppo.storage
    out = dict(n=n, t=t, d_obs=d_obs, d_act=d_act, obs=obs.numpy(), rewards=rewards.numpy(), dones=dones.numpy().astype(np.uint8),
               time_outs=time_outs.numpy().astype(np.uint8), actions=np.stack(actions),
               st_rewards=st.rewards.numpy().reshape(t, n).copy(), st_values=st.values.numpy().reshape(t, n).copy(),
               st_log_prob=st.actions_log_prob.numpy().reshape(t, n).copy(), st_returns=st.returns.numpy().reshape(t, n).copy(),
               st_advantages=st.advantages.numpy().reshape(t, n).copy(), st_mu=st.mu.numpy().copy(), st_sigma=st.sigma.numpy().copy(),
               **{"cfg/" + k: (v if not isinstance(v, str) else np.array(v)) for k, v in cfgs.items()})
    for k, v in init.items():
        out["init/" + k] = v
    torch.manual_seed(29)
    out["indices"] = torch.randperm(cfgs["num_mini_batches"] * (n * t // cfgs["num_mini_batches"])).numpy()  # the generator's first draw (:159)
    torch.manual_seed(29)
    mvl, msl = ppo.update()
    out["mean_value_loss"], out["mean_surrogate_loss"] = np.float64(mvl), np.float64(msl)
    for k, v in ac.state_dict().items():
        out["final/" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(GOLDEN, "F13_rsl_ppo.npz"), **out)
    print("F13_rsl_ppo saved: value loss", mvl, "surrogate", msl, "bootstrapped", int(time_outs.sum()), "dones", int(dones.sum()))


if __name__ == "__main__":
    main()
