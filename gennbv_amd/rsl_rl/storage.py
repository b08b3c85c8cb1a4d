"""RolloutStorage (rsl_rl/storage/rollout_storage.py:37-192, feed-forward policies).

Same public surface -- `Transition`, `add_transitions`, `clear`, `compute_returns`, `get_statistics`,
`mini_batch_generator`, the tensor attributes -- with the return / advantage computation on the GAE kernel
(`csrc/gae.hip: k_gae<true>`: the reference's reverse Python loop of T x 8 launches is one launch, followed by the
whole-buffer advantage normalisation of :143-144).  rsl_rl's mask convention: `dones[t]` is stored with the transition
it ended, the scan multiplies by `1 - dones[t]` (:137-140)."""
from __future__ import annotations

import torch

from .. import gae as gae_ops


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.observations = None
            self.critic_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None
            self.hidden_states = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        self.device = device
        self.obs_shape, self.privileged_obs_shape, self.actions_shape = obs_shape, privileged_obs_shape, actions_shape
        t, n = num_transitions_per_env, num_envs
        z = lambda *s: torch.zeros(t, n, *s, device=device)  # noqa: E731
        self.observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape) if privileged_obs_shape[0] is not None else None
        self.rewards, self.actions, self.dones = z(1), z(*actions_shape), z(1).byte()
        self.actions_log_prob, self.values, self.returns, self.advantages = z(1), z(1), z(1), z(1)
        self.mu, self.sigma = z(*actions_shape), z(*actions_shape)
        self.num_transitions_per_env, self.num_envs = t, n
        self.step = 0

    def add_transitions(self, transition: "RolloutStorage.Transition"):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        s = self.step
        self.observations[s].copy_(transition.observations)
        if self.privileged_observations is not None:
            self.privileged_observations[s].copy_(transition.critic_observations)
        self.actions[s].copy_(transition.actions)
        self.rewards[s].copy_(transition.rewards.view(-1, 1))
        self.dones[s].copy_(transition.dones.view(-1, 1))
        self.values[s].copy_(transition.values)
        self.actions_log_prob[s].copy_(transition.actions_log_prob.view(-1, 1))
        self.mu[s].copy_(transition.action_mean)
        self.sigma[s].copy_(transition.action_sigma)
        if transition.hidden_states not in (None, (None, None)):
            raise NotImplementedError("recurrent policies are not on the GenNBV path")
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        """:130-144 -- returns by the reverse scan, advantages = returns - values normalised over the whole buffer."""
        ret, adv = gae_ops.compute_returns_rsl(self.rewards, self.values, self.dones, last_values, gamma, lam, normalize=True)
        self.returns.copy_(ret.view_as(self.returns))
        self.advantages = adv.view_as(self.returns)

    def get_statistics(self):
        """:146-154: (mean trajectory length, mean reward) of the stored rollout; a trajectory ends at a done flag or at the buffer's
        last row (the reference forces `dones[-1] = 1` IN the storage: kept, callers see it)."""
        self.dones[-1] = 1
        ends = self.dones.transpose(0, 1).reshape(-1).nonzero().flatten()  # env-major positions of the trajectory ends
        lengths = torch.diff(ends, prepend=ends.new_full((1,), -1))
        return lengths.float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8, indices=None):
        """:156-192.  `indices` (additive): a fixed permutation instead of `torch.randperm` on the storage's device (the
        device generators of different back ends do not share a random stream: parity tests pass the reference's draw)."""
        batch_size = self.num_envs * self.num_transitions_per_env
        mini_batch_size = batch_size // num_mini_batches
        if indices is None:
            indices = torch.randperm(num_mini_batches * mini_batch_size, requires_grad=False, device=self.device)
        obs = self.observations.flatten(0, 1)
        critic_obs = self.privileged_observations.flatten(0, 1) if self.privileged_observations is not None else obs
        actions, values, returns = self.actions.flatten(0, 1), self.values.flatten(0, 1), self.returns.flatten(0, 1)
        old_lp, adv = self.actions_log_prob.flatten(0, 1), self.advantages.flatten(0, 1)
        old_mu, old_sigma = self.mu.flatten(0, 1), self.sigma.flatten(0, 1)
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                b = indices[i * mini_batch_size:(i + 1) * mini_batch_size]
                yield (obs[b], critic_obs[b], actions[b], values[b], adv[b], returns[b], old_lp[b], old_mu[b], old_sigma[b],
                       (None, None), None)
