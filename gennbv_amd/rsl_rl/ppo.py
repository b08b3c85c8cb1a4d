"""PPO (rsl_rl/algorithms/ppo.py:39-199): `act / process_env_step / compute_returns / update` with the reference's
signatures over any feed-forward `actor_critic` that offers rsl_rl's module protocol (`act`, `evaluate`,
`get_actions_log_prob`, `action_mean`, `action_std`, `entropy`, `reset`, `is_recurrent = False`).

On the GPU the scalar loss and its gradient are one kernel (`gnbv_ppo_loss_rsl`), clip_grad_norm_ + Adam run over one flat
buffer (`gnbv_clip_adam_step`, torch.optim.Adam's default eps 1e-8), the return scan is `k_gae<true>`; the running loss
means are device-side sums read once per update() (the reference reads two `.item()` per minibatch, :187-188).
The module forward / backward stays with torch autograd: rsl_rl's ActorCritic is a small Gaussian MLP that GenNBV never
trains (its policy lives on the stable_baselines3 path)."""
from __future__ import annotations

import torch

from .. import _lib
from .storage import RolloutStorage


class _FlatAdamView:
    """`alg.optimizer` as the reference's runner uses it (rsl_rl/runners/on_policy_runner.py: `optimizer.state_dict()` /
    `load_state_dict()` in save / load, `param_groups[...]["lr"]`): a torch.optim.Adam whose state lives in the flat HIP optimizer
    once the first update() created it.  Before that it IS the torch optimizer's state."""

    def __init__(self, owner, torch_adam):
        self._owner, self._adam = owner, torch_adam

    def __getattr__(self, name):  # param_groups, defaults, ...
        if name in ("_adam", "_owner") or (name.startswith("__") and name.endswith("__")):
            raise AttributeError(name)  # (copy.deepcopy / pickle probe the instance before __init__ ran: no recursion through _adam)
        return getattr(self._adam, name)

    def state_dict(self):
        flat = self._owner._flat
        return flat.torch_state_dict(self._adam) if flat is not None else self._adam.state_dict()

    def load_state_dict(self, sd):
        self._adam.load_state_dict(sd)
        flat = self._owner._flat
        if flat is not None:
            flat.load_torch_state_dict(self._adam.state_dict())
            for g in self._adam.param_groups:
                flat.lr = float(g["lr"])
        if self._owner.schedule != "adaptive":  # (the reference's adaptive schedule restarts from the CONFIGURED rate after a load:
            for g in self._adam.param_groups:   # rsl_rl/algorithms/ppo.py:148-159 writes `self.learning_rate` into the groups, never reads them)
                self._owner.learning_rate = float(g["lr"])


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
                 schedule="fixed", desired_kl=0.01, device="cpu"):
        self.device = device
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic
        self.actor_critic.to(self.device)
        self.storage = None
        self._flat = None
        self.optimizer = _FlatAdamView(self, torch.optim.Adam(self.actor_critic.parameters(), lr=learning_rate))
        # (self._flat: ops.ppo_ops.FlatAdam, created at the first update() on a GPU; `optimizer` reads / writes its state from then on)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam, self.max_grad_norm = gamma, lam, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device)

    def test_mode(self):
        self.actor_critic.test()

    def train_mode(self):
        self.actor_critic.train()

    def act(self, obs, critic_obs):
        if getattr(self.actor_critic, "is_recurrent", False):
            raise NotImplementedError("recurrent policies are not on the GenNBV path")
        t = self.transition
        t.actions = self.actor_critic.act(obs).detach()
        t.values = self.actor_critic.evaluate(critic_obs).detach()
        t.actions_log_prob = self.actor_critic.get_actions_log_prob(t.actions).detach()
        t.action_mean = self.actor_critic.action_mean.detach()
        t.action_sigma = self.actor_critic.action_std.detach()
        t.observations, t.critic_observations = obs, critic_obs  # (recorded before env.step())
        return t.actions

    def process_env_step(self, rewards, dones, infos):
        t = self.transition
        t.rewards = rewards.clone()
        t.dones = dones
        if "time_outs" in infos:  # bootstrapping on time outs with V(obs of the acting step) (:112-116)
            t.rewards += self.gamma * torch.squeeze(t.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    def update(self, indices=None):
        """:127-199.  -> (mean_value_loss, mean_surrogate_loss).  `indices`: see RolloutStorage.mini_batch_generator."""
        lib = _lib.load()
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise _lib.GennbvHipError("gennbv_amd.rsl_rl.PPO.update runs on the GPU only (no CPU fallback)")
        if self._flat is None:
            from ..ops.ppo_ops import FlatAdam
            eps, betas = self.optimizer.defaults["eps"], self.optimizer.defaults["betas"]
            self._flat = FlatAdam(self.actor_critic, lr=self.learning_rate, betas=betas, eps=eps)
            self._flat.load_torch_adam_state(self.optimizer._adam)
        opt = self._flat
        sums = torch.zeros(2, dtype=torch.float32, device=dev)
        st = lambda: _lib.stream_ptr(dev)  # noqa: E731
        batches = self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, indices=indices)
        for (obs_b, critic_obs_b, actions_b, target_values_b, adv_b, returns_b, old_lp_b, old_mu_b, old_sigma_b, _hid, _masks) in batches:
            self.actor_critic.act(obs_b)
            lp = self.actor_critic.get_actions_log_prob(actions_b)
            value = self.actor_critic.evaluate(critic_obs_b)
            mu_b, sigma_b, entropy = self.actor_critic.action_mean, self.actor_critic.action_std, self.actor_critic.entropy
            if self.desired_kl is not None and self.schedule == "adaptive":
                with torch.inference_mode():
                    kl = torch.sum(torch.log(sigma_b / old_sigma_b + 1.e-5) + (torch.square(old_sigma_b) + torch.square(old_mu_b - mu_b))
                                   / (2.0 * torch.square(sigma_b)) - 0.5, axis=-1)
                    kl_mean = float(torch.mean(kl))
                if kl_mean > self.desired_kl * 2.0:
                    self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                elif self.desired_kl / 2.0 > kl_mean > 0.0:
                    self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                for g in self.optimizer.param_groups:
                    g["lr"] = self.learning_rate
            opt.lr = self.learning_rate
            b = int(lp.shape[0])
            flat = lambda x: x.detach().reshape(-1).contiguous().float()  # noqa: E731
            lp_c, v_c = lp.reshape(-1).contiguous(), value.reshape(-1).contiguous()
            ent_c = entropy.reshape(-1).contiguous()
            d_lp, d_v, d_ent = torch.empty_like(lp_c), torch.empty_like(v_c), torch.empty_like(ent_c)
            olp, adv, tv, ret = flat(old_lp_b), flat(adv_b), flat(target_values_b), flat(returns_b)
            _lib.check(lib.gnbv_ppo_loss_rsl(b, lp_c.data_ptr(), olp.data_ptr(), adv.data_ptr(), v_c.data_ptr(), tv.data_ptr(), ret.data_ptr(),
                                             float(self.clip_param), float(self.value_loss_coef), float(self.entropy_coef),
                                             int(bool(self.use_clipped_value_loss)), d_lp.data_ptr(), d_v.data_ptr(), d_ent.data_ptr(),
                                             sums.data_ptr(), st()), "gnbv_ppo_loss_rsl")
            opt.zero_grad()
            torch.autograd.backward([lp_c, v_c, ent_c], [d_lp, d_v, d_ent])
            opt.step(self.max_grad_norm)
        num_updates = self.num_learning_epochs * self.num_mini_batches
        s = (sums / num_updates).cpu()  # the ONE read-back of update()
        self.storage.clear()
        return float(s[0]), float(s[1])
