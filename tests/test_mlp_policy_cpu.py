"""CPU: BASELINE configs[0] -- 4 envs, 16^3 grid, MLP policy, PPO on the CPU -- against fixture F15_ppo_mlp_c0, which the REFERENCE's own
ActorCriticPolicy_Train_Eval (FlattenExtractor + MlpExtractor: stable_baselines3/common/policies.py:844,868-872, torch_layers.py:34-46,
135-240) and PPO_Grid_Obs.train() (ppo/ppo_grid_obs.py:176-297) produced (oracle/gen_golden_mlp.py).  Two architectures: SB3's default for
an MLP policy (net_arch=None -> two towers of 64, 64) and one with a shared layer and towers of different depth."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests import policy_util as pu

ARCHS = {"default": None, "shared": [96, dict(pi=[48], vf=[32, 16])]}
TOL = 1e-5


def _obs(fx):
    return torch.from_numpy(np.concatenate([fx["obs_state"], fx["obs_grid"].astype(np.float32), fx["obs_rgb_u8"].astype(np.float32) / np.float32(255.0)], axis=1))


def _ppo(fx, tag):
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    t, n, g = int(fx["T"]), int(fx["N"]), int(fx["G"])
    obs_space, act_space = pu.spaces(g)

    class _Env:
        num_envs, device, max_episode_length = n, "cpu", 100
        observation_space, action_space = obs_space, act_space
    kw = {} if ARCHS[tag] is None else {"net_arch": ARCHS[tag]}
    ppo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, _Env(), learning_rate=float(fx["lr"]), n_steps=t, batch_size=int(fx["batch_size"]),
                       n_epochs=int(fx["n_epochs"]), gamma=0.99, gae_lambda=0.95, clip_range=0.2, clip_range_vf=0.2, ent_coef=0.01, vf_coef=0.8,
                       max_grad_norm=1.0, target_kl=None, device="cpu", policy_kwargs=kw)
    sd = ppo.policy.state_dict()
    assert list(sd.keys()) == [str(k) for k in fx[tag + "/sd_names"]], "the reference's module tree / checkpoint keys"
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in fx[tag + "/sd_shapes"]]
    ppo.policy.load_state_dict({k: torch.from_numpy(v) for k, v in gu.det_state_dict({k: tuple(v.shape) for k, v in sd.items()}).items()})
    return ppo


@pytest.mark.parametrize("tag", list(ARCHS))
def test_mlp_policy_forward_vs_reference(tag):
    fx = gu.load("F15_ppo_mlp_c0")
    ppo = _ppo(fx, tag)
    pol = ppo.policy
    from gennbv_amd.sb3.torch_layers import FlattenExtractor, MlpExtractor
    assert isinstance(pol.features_extractor, FlattenExtractor) and isinstance(pol.mlp_extractor, MlpExtractor)
    assert pol.features_dim == 600 + 16 ** 3 + 8192
    obs, actions = _obs(fx), torch.from_numpy(fx["actions"]).view(-1, 6)
    pol.set_training_mode(False)
    with torch.no_grad():
        values, log_prob, entropy = pol.evaluate_actions(obs, actions)
        logits = pol.action_net(pol.mlp_extractor.forward_actor(pol.extract_features(obs)))
        a, v2, lp2 = pol(obs, deterministic=True)
    np.testing.assert_allclose(values.numpy(), fx[tag + "/eval_values"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(logits.numpy(), fx[tag + "/eval_logits"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(log_prob.numpy(), fx[tag + "/eval_log_prob"], rtol=TOL, atol=1e-4)
    np.testing.assert_allclose(entropy.numpy(), fx[tag + "/eval_entropy"], rtol=TOL, atol=1e-4)
    np.testing.assert_allclose(v2.numpy(), fx[tag + "/eval_values"], rtol=TOL, atol=TOL)
    assert a.shape == (obs.shape[0], 6) and torch.equal(pol.predict_values(obs), v2)


@pytest.mark.parametrize("tag", list(ARCHS))
def test_mlp_policy_ppo_train_vs_reference(tag):
    fx = gu.load("F15_ppo_mlp_c0")
    ppo = _ppo(fx, tag)
    t, n = int(fx["T"]), int(fx["N"])
    buf = ppo.rollout_buffer
    buf.observations[:t].copy_(_obs(fx).view(t, n, -1))
    buf.actions.copy_(torch.from_numpy(fx["actions"]))
    buf.values.copy_(torch.from_numpy(fx[tag + "/values"]))
    buf.log_probs.copy_(torch.from_numpy(fx[tag + "/log_probs"]).view(t, n, 1))
    buf.rewards.copy_(torch.from_numpy(fx["rewards"]).view(t, n, 1))
    buf.advantages.copy_(torch.from_numpy(fx[tag + "/advantages"]).view(t, n, 1))
    buf.returns.copy_(torch.from_numpy(fx[tag + "/returns"]).view(t, n, 1))
    buf.step = t
    buf.indices = fx["indices"].copy()
    buf._indices_dev = None
    traj = []
    orig = ppo.policy.optimizer.step

    def hook(*a, **k):
        r = orig(*a, **k)
        traj.append([float(p.detach().double().sum()) for p in ppo.policy.parameters()])
        return r
    ppo.policy.optimizer.step = hook
    ppo.train()
    assert len(traj) == int(fx[tag + "/n_optimizer_steps"]) == 12
    log = ppo.logger.name_to_value
    for k in ("train/entropy_loss", "train/policy_gradient_loss", "train/value_loss", "train/approx_kl", "train/clip_fraction", "train/loss",
              "train/explained_variance"):
        ref = float(fx[tag + "/log/" + k])
        assert abs(float(log[k]) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, log[k], ref)  # north_star: PPO loss within 1e-4
    np.testing.assert_allclose(np.array(traj), fx[tag + "/param_sum_trajectory"], rtol=2e-4, atol=2e-3)
    for name, p in ppo.policy.named_parameters():
        a = p.detach().numpy()
        mine = a if a.size <= 70000 else a.reshape(-1)[::97]
        np.testing.assert_allclose(mine, fx[tag + "/final/" + name], rtol=1e-3, atol=2e-4, err_msg=name)


def test_mlp_policy_learn_over_a_recorded_feed_on_the_cpu():
    """configs[0] end to end: `learn()` -- collect_rollouts + GAE + train(), the plain-torch path -- with SB3's default MLP policy over the
    observations the REFERENCE env produced from a recorded depth / pose feed (fixture F11, tests/rollout_util.RecordedEnv: no simulator, no
    GPU, no kernels).  Two rollouts of the recording: every step consumed, parameters moved and finite, the reference's logger keys
    present."""
    from tests import rollout_util as ru
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    from gennbv_amd.sb3.torch_layers import MlpExtractor
    fx = gu.load("F11_rollout")
    t, n = int(fx["T"]), int(fx["n"])
    env = ru.RecordedEnv(fx, check_actions=False)  # (an MLP policy samples other actions than the recording's policy: the replay ignores them)
    torch.manual_seed(3)
    np.random.seed(3)
    algo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, env, learning_rate=1e-5, n_steps=t, batch_size=8, n_epochs=2, gamma=0.99, gae_lambda=0.95,
                        device="cpu", seed=None)
    assert isinstance(algo.policy.mlp_extractor, MlpExtractor) and algo.policy.mlp_extractor.latent_dim_pi == 64
    before = torch.cat([p.detach().reshape(-1)[:1000] for p in algo.policy.parameters()]).clone()
    algo.learn(total_timesteps=2 * t * n)
    assert algo.num_timesteps == 2 * t * n and env.k == 2 * t, "two rollouts of the recording, step by step"
    after = torch.cat([p.detach().reshape(-1)[:1000] for p in algo.policy.parameters()])
    assert bool(torch.isfinite(after).all()) and float((after - before).abs().max()) > 0
    log = algo.logger.name_to_value
    for k in ("train/loss", "train/approx_kl", "train/value_loss", "train/entropy_loss", "train/policy_gradient_loss", "train/n_updates"):
        assert k in log and np.isfinite(float(log[k])), k
    assert int(log["train/n_updates"]) == 4


def test_cpu_placed_buffer_gae_equals_the_reference_fixture():
    """TensorRolloutBuffer_Grid_Obs on the CPU (configs[0]) computes advantages / returns with its torch recurrence: bit-identical to
    fixture F8, the reference's own compute_returns_and_advantage (buffers.py:706-724)."""
    from gennbv_amd.sb3.buffers import TensorRolloutBuffer_Grid_Obs
    from gennbv_amd.spaces import Box, MultiDiscrete
    fx = gu.load("F8_gae")
    t, n = fx["rewards"].shape[:2]
    np.random.seed(0)
    buf = TensorRolloutBuffer_Grid_Obs(t, Box(-1, 1, shape=(3,)), MultiDiscrete([3]), device="cpu", gamma=0.99, gae_lambda=0.95, n_envs=n)
    buf.rewards.copy_(torch.from_numpy(fx["rewards"]).view(t, n, 1))
    buf.values.copy_(torch.from_numpy(fx["values"]).view(t, n, 1))
    buf.episode_starts.copy_(torch.from_numpy(fx["episode_starts"]).view(buf.episode_starts.shape).to(buf.episode_starts.dtype))
    buf.compute_returns_and_advantage(torch.from_numpy(fx["last_values"]).view(n, 1), torch.from_numpy(fx["dones"].astype(np.int64)))
    assert buf.advantages.view(t, n).numpy().tobytes() == fx["sb3_advantages"].tobytes()
    assert buf.returns.view(t, n).numpy().tobytes() == fx["sb3_returns"].tobytes()


def test_mlp_extractor_seeded_initialisation_equals_the_reference():
    # F16 = the reference's own MlpExtractor constructed behind torch.manual_seed(3): the default (non-orthogonal) initialisation
    # consumes the global generator in layer-CREATION order -- policy layer i, then value layer i, depth by depth
    from gennbv_amd.sb3.torch_layers import MlpExtractor
    fx = gu.load("F16_mlp_init")
    archs = {"default": [dict(pi=[64, 64], vf=[64, 64])], "shared": [96, dict(pi=[48], vf=[32, 16])], "ragged": [32, dict(pi=[8, 9, 10], vf=[7])]}
    for tag, arch in archs.items():
        torch.manual_seed(int(fx["seed"]))
        m = MlpExtractor(int(fx["feature_dim"]), arch, torch.nn.Tanh, "cpu")
        sd = m.state_dict()
        ref_keys = [k.split("/", 1)[1] for k in fx.files if k.startswith(tag + "/")]
        assert list(sd) == ref_keys, tag
        for k in ref_keys:
            assert np.array_equal(sd[k].numpy(), fx[f"{tag}/{k}"]), (tag, k)
