"""GPU: the rsl_rl flavour (gennbv_amd/rsl_rl: RolloutStorage + PPO on k_gae<true>, gnbv_ppo_loss_rsl, the flat clip + Adam)
against the vendored rsl_rl's own PPO run by oracle/gen_golden_rsl.py (fixture F13): the storage after a rollout with
time-out bootstraps, returns / normalised advantages, the mean losses of update() and the parameters after 6 steps.
The sampled actions and the minibatch permutation of the reference are fed in (different back-end random streams)."""
import numpy as np
import pytest
import torch
from torch import nn
from torch.distributions import Normal

from tests import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _ActorCritic(nn.Module):
    """rsl_rl's module protocol over a small Gaussian MLP (same layers / state_dict keys as the fixture's network)."""
    is_recurrent = False

    def __init__(self, d_obs, d_act, forced_actions):
        super().__init__()
        mlp = lambda o: nn.Sequential(nn.Linear(d_obs, 32), nn.ELU(), nn.Linear(32, 32), nn.ELU(), nn.Linear(32, o))  # noqa: E731
        self.actor, self.critic = mlp(d_act), mlp(1)
        self.std = nn.Parameter(torch.ones(d_act))
        self.distribution = None
        self._forced = list(forced_actions)

    def reset(self, dones=None):
        pass

    action_mean = property(lambda self: self.distribution.mean)
    action_std = property(lambda self: self.distribution.stddev)
    entropy = property(lambda self: self.distribution.entropy().sum(dim=-1))

    def act(self, observations, **kw):
        mean = self.actor(observations)
        self.distribution = Normal(mean, mean * 0. + self.std)
        return self._forced.pop(0) if (self._forced and not torch.is_grad_enabled()) else self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def evaluate(self, critic_observations, **kw):
        return self.critic(critic_observations)


def test_rsl_rl_ppo_matches_the_vendored_reference():
    from gennbv_amd.rsl_rl import PPO
    fx = gu.load("F13_rsl_ppo")
    n, t, d_obs, d_act = (int(fx[k]) for k in ("n", "t", "d_obs", "d_act"))
    cfg = {k[4:]: (str(fx[k]) if fx[k].dtype.kind == "U" else fx[k].item()) for k in fx.files if k.startswith("cfg/")}
    ac = _ActorCritic(d_obs, d_act, [torch.from_numpy(a).to(DEV) for a in fx["actions"]])
    ac.load_state_dict({k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("init/")})
    ppo = PPO(ac, device=DEV, **cfg)
    ppo.init_storage(n, t, [d_obs], [None], [d_act])
    obs = torch.from_numpy(fx["obs"]).to(DEV)
    with torch.no_grad():
        for s in range(t):
            a = ppo.act(obs[s], obs[s])
            assert torch.equal(a.cpu(), torch.from_numpy(fx["actions"][s]))
            ppo.process_env_step(torch.from_numpy(fx["rewards"][s]).to(DEV), torch.from_numpy(fx["dones"][s].astype(bool)).to(DEV),
                                 {"time_outs": torch.from_numpy(fx["time_outs"][s].astype(bool)).to(DEV)})
        ppo.compute_returns(obs[t])
    st = ppo.storage
    f = lambda x: x.detach().cpu().numpy().reshape(t, n)  # noqa: E731
    np.testing.assert_allclose(f(st.values), fx["st_values"], atol=2e-6)
    np.testing.assert_allclose(f(st.actions_log_prob), fx["st_log_prob"], atol=5e-6)
    boot = fx["time_outs"].astype(bool)
    assert np.array_equal(f(st.rewards)[~boot], fx["rewards"][~boot])  # untouched where no time-out
    np.testing.assert_allclose(f(st.rewards), fx["st_rewards"], atol=2e-6)  # + gamma * V(obs of the acting step) elsewhere
    assert boot.sum() > 0 and np.abs(fx["st_rewards"] - fx["rewards"])[boot].min() > 1e-4
    np.testing.assert_allclose(f(st.returns), fx["st_returns"], atol=1e-5)
    np.testing.assert_allclose(f(st.advantages), fx["st_advantages"], atol=1e-5)
    mvl, msl = ppo.update(indices=torch.from_numpy(fx["indices"]).to(DEV))
    assert abs(mvl - float(fx["mean_value_loss"])) < 1e-5 * max(1.0, abs(float(fx["mean_value_loss"])))
    assert abs(msl - float(fx["mean_surrogate_loss"])) < 1e-5
    for k, v in ac.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), fx["final/" + k], rtol=1e-4, atol=2e-5, err_msg=k)
    assert st.step == 0  # storage.clear()
    # `alg.optimizer` (what the reference's runner saves / loads, on_policy_runner.py) carries the flat optimizer's state ...
    sd = ppo.optimizer.state_dict()
    n_up = int(cfg["num_learning_epochs"]) * int(cfg["num_mini_batches"])
    ps = list(ac.parameters())
    assert len(sd["state"]) == len(ps) and all(int(e["step"]) == n_up for e in sd["state"].values())
    assert all(float(e["exp_avg_sq"].abs().sum()) > 0 for e in sd["state"].values())
    assert abs(sd["param_groups"][0]["lr"] - ppo.learning_rate) < 1e-12
    # ... and a second algorithm object that loads it into a fresh torch Adam owns the same moments after its first update() builds
    # the flat optimizer (load -> FlatAdam.load_torch_adam_state), as does one that loads it after that
    ac2 = _ActorCritic(d_obs, d_act, [])
    ac2.load_state_dict(ac.state_dict())
    ppo2 = PPO(ac2, device=DEV, **cfg)
    ppo2.optimizer.load_state_dict(sd)
    sd2 = ppo2.optimizer.state_dict()
    for i in sd["state"]:
        assert torch.equal(sd2["state"][i]["exp_avg"].cpu(), sd["state"][i]["exp_avg"].cpu())
    ppo.optimizer.load_state_dict(sd2)  # (the flat optimizer exists: written through)
    sd3 = ppo.optimizer.state_dict()
    for i in sd["state"]:
        assert torch.equal(sd3["state"][i]["exp_avg_sq"].cpu(), sd["state"][i]["exp_avg_sq"].cpu()) and int(sd3["state"][i]["step"]) == n_up


def test_rsl_loss_kernel_gradients_vs_torch_autograd():
    """gnbv_ppo_loss_rsl against torch autograd on the reference's expression (:160-180), both value-loss modes."""
    import ctypes  # noqa: F401
    from gennbv_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    B = 300
    for clipped in (1, 0):
        lp = (torch.randn(B, generator=g) * 0.3).to(DEV).requires_grad_(True)
        olp = (torch.randn(B, generator=g) * 0.3).to(DEV)
        adv, v = torch.randn(B, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV).requires_grad_(True)
        tv, ret = torch.randn(B, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV)
        ent = torch.rand(B, generator=g).to(DEV).requires_grad_(True)
        ratio = torch.exp(lp - olp)
        sur = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 0.8, 1.2)).mean()
        if clipped:
            vc = tv + (v - tv).clamp(-0.2, 0.2)
            vl = torch.max((v - ret).pow(2), (vc - ret).pow(2)).mean()
        else:
            vl = (ret - v).pow(2).mean()
        (sur + 0.7 * vl - 0.01 * ent.mean()).backward()
        d = [torch.empty(B, device=DEV) for _ in range(3)]
        sums = torch.zeros(2, device=DEV)
        _lib.check(lib.gnbv_ppo_loss_rsl(B, lp.data_ptr(), olp.data_ptr(), adv.data_ptr(), v.data_ptr(), tv.data_ptr(), ret.data_ptr(), 0.2, 0.7,
                                         0.01, clipped, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), sums.data_ptr(),
                                         _lib.stream_ptr(torch.device(DEV))), "gnbv_ppo_loss_rsl")
        for got, want in zip(d, (lp.grad, v.grad, ent.grad)):
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-7)
        assert abs(float(sums[0]) - float(vl)) < 1e-5 and abs(float(sums[1]) - float(sur)) < 1e-5
