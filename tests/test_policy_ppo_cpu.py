"""CPU: policy / PPO host logic (torch fp32 reference path) against the goldens produced by the
reference's own Hybrid_Encoder / ActorCriticPolicy_Train_Eval / PPO_Grid_Obs.train()."""
import hashlib

import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests import policy_util as pu

TOL = 1e-5  # fp32 forward tolerance; same torch build on this host => near bit-exact


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_state_dict_keys_and_det_weights_match_reference():
    fx = gu.load("F7_policy")
    pol, _, _ = pu.make_policy()
    sd = pol.state_dict()
    assert list(sd.keys()) == [str(k) for k in fx["sd_names"]]  # reference checkpoint format
    assert [sha(v.numpy()) for v in sd.values()] == [str(s) for s in fx["sd_sha"]]
    assert sum(p.numel() for p in pol.parameters()) == 1143553


def test_policy_forward_eval_and_train_mode_vs_reference():
    fx = gu.load("F7_policy")
    pol, _, _ = pu.make_policy()
    obs, actions = pu.unpack_obs(fx), torch.from_numpy(fx["actions"])
    pol.set_training_mode(False)
    with torch.no_grad():
        feats = pol.extract_features(obs)
        values, log_prob, entropy = pol.evaluate_actions(obs, actions)
        logits = pol.action_net(feats)
    np.testing.assert_allclose(feats.numpy(), fx["eval_features"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(values.numpy(), fx["eval_values"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(logits.numpy(), fx["eval_logits"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(log_prob.numpy(), fx["eval_log_prob"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(entropy.numpy(), fx["eval_entropy"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(pol.predict_values(obs).detach().numpy(), fx["eval_predict_values"], rtol=TOL, atol=TOL)
    pol.set_training_mode(True)
    pol.zero_grad()
    values, log_prob, entropy = pol.evaluate_actions(obs, actions)
    b = obs.shape[0]
    w = torch.linspace(0.5, 1.5, b)
    loss = (values.flatten() * w).sum() + (log_prob * w.flip(0)).sum() + 0.3 * (entropy * w).sum()
    loss.backward()
    np.testing.assert_allclose(values.detach().numpy(), fx["train_values"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(log_prob.detach().numpy(), fx["train_log_prob"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(entropy.detach().numpy(), fx["train_entropy"], rtol=TOL, atol=TOL)
    for name, p in pol.named_parameters():
        g = p.grad.numpy()
        ref = fx["grad/" + name]
        mine = g if g.size <= 70000 else g.reshape(-1)[::97]
        np.testing.assert_allclose(mine, ref, rtol=1e-4, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), fx["grad_norm/" + name], rtol=1e-4, atol=1e-4)
    for k, v in pol.state_dict().items():
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(v.numpy(), fx["bn_after/" + k], rtol=1e-6, atol=1e-6)


def _ppo_from_fixture(fx, device="cpu", backend="torch", **enc_kw):
    from gennbv_amd.sb3.ppo_grid_obs import PPO_Grid_Obs
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    from tests.torch_reference import encoder_class
    t, n = int(fx["T"]), int(fx["N"])
    _, obs_space, act_space = pu.make_policy(det_weights=False)

    class _Env:
        num_envs = n
        observation_space, action_space = obs_space, act_space
        max_episode_length = 100
    _Env.device = device
    tkl = float(fx["target_kl"])
    ppo = PPO_Grid_Obs(ActorCriticPolicy_Train_Eval, _Env(), learning_rate=float(fx["lr"]), n_steps=t,
                       batch_size=int(fx["batch_size"]), n_epochs=int(fx["n_epochs"]), gamma=0.99, gae_lambda=0.95,
                       clip_range=0.2, clip_range_vf=0.2, ent_coef=0.01, vf_coef=0.8, max_grad_norm=1.0,
                       target_kl=None if tkl < 0 else tkl, device=device,
                       policy_kwargs=dict(net_arch=[], features_extractor_class=encoder_class(backend),
                                          features_extractor_kwargs=dict(
                                              encoder_param={"hidden_shapes": [256, 256], "visual_dim": 256},
                                              net_param={"transformer_params": [[1, 256], [1, 256]],
                                                         "append_hidden_shapes": [256, 256]},
                                              state_input_shape=(600,), visual_input_shape=(100, 400, 400),
                                              grid_size=20, **enc_kw)))
    shapes = {k: tuple(v.shape) for k, v in ppo.policy.state_dict().items()}
    ppo.policy.load_state_dict({k: torch.from_numpy(v).to(device) for k, v in gu.det_state_dict(shapes).items()})
    buf = ppo.rollout_buffer
    obs = pu.unpack_obs(fx).view(t, n, -1).to(device)
    buf.observations[:t].copy_(obs)
    buf.actions.copy_(torch.from_numpy(fx["actions"]).to(device))
    buf.values.copy_(torch.from_numpy(fx["values"]).to(device))
    buf.log_probs.copy_(torch.from_numpy(fx["log_probs"]).view(t, n, 1).to(device))
    buf.rewards.copy_(torch.from_numpy(fx["rewards"]).view(t, n, 1).to(device))
    buf.advantages.copy_(torch.from_numpy(fx["advantages"]).view(t, n, 1).to(device))
    buf.returns.copy_(torch.from_numpy(fx["returns"]).view(t, n, 1).to(device))
    buf.step = t
    buf.indices = fx["indices"].copy()
    buf._indices_dev = None
    return ppo


def check_ppo_against_fixture(ppo, fx, loss_tol=1e-4, param_rtol=2e-4):
    traj = []
    orig = ppo.policy.optimizer.step

    def hook(*a, **k):
        r = orig(*a, **k)
        traj.append([float(p.detach().double().sum()) for p in ppo.policy.parameters()])
        return r
    ppo.policy.optimizer.step = hook
    ppo.train()
    assert len(traj) == int(fx["n_optimizer_steps"])  # early-stop position
    log = ppo.logger.name_to_value
    for k in ("train/entropy_loss", "train/policy_gradient_loss", "train/value_loss", "train/approx_kl",
              "train/clip_fraction", "train/loss", "train/explained_variance"):
        assert abs(float(log[k]) - float(fx["log/" + k])) <= loss_tol * max(1.0, abs(float(fx["log/" + k]))), (k, log[k], fx["log/" + k])
    if len(traj):
        np.testing.assert_allclose(np.array(traj), fx["param_sum_trajectory"], rtol=param_rtol, atol=2e-3)
    for name, p in ppo.policy.named_parameters():
        a = p.detach().cpu().numpy()
        mine = a if a.size <= 70000 else a.reshape(-1)[::97]
        np.testing.assert_allclose(mine, fx["final/" + name], rtol=1e-3, atol=2e-4, err_msg=name)
    for k, v in ppo.policy.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), fx["final_bn/" + k], rtol=1e-4, atol=1e-5)
        if "num_batches" in k:
            assert int(v) == int(fx["final_bn/" + k])


@pytest.mark.parametrize("name", ["F9_ppo_train", "F9_ppo_train_earlystop"])
def test_ppo_train_matches_reference_on_recorded_rollout(name):
    fx = gu.load(name)
    torch.manual_seed(0)
    ppo = _ppo_from_fixture(fx)
    check_ppo_against_fixture(ppo, fx)


def test_minibatch_rows_equal_swap_and_flatten_order():
    """C3: index i of the reference's swapped/flattened buffer is (env i // T, step i % T)."""
    from gennbv_amd.sb3.buffers import TensorRolloutBuffer_Grid_Obs
    from gennbv_amd.spaces import Box, MultiDiscrete
    t, n, d = 5, 3, 7
    np.random.seed(4)
    buf = TensorRolloutBuffer_Grid_Obs(t, Box(-1, 1, shape=(d,)), MultiDiscrete([3, 4]), device="cpu", n_envs=n)
    obs = torch.arange(t * n * d, dtype=torch.float32).view(t, n, d)
    buf.observations[:t].copy_(obs)
    buf.values.copy_(torch.arange(t * n, dtype=torch.float32).view(t, n, 1))
    buf.step = t
    ref_flat = obs.swapaxes(0, 1).reshape(t * n, d)
    got = torch.cat([mb.observations for mb in buf.get(4)])
    assert torch.equal(got, ref_flat[torch.from_numpy(buf.indices)])
    v_ref = buf.values.swapaxes(0, 1).reshape(-1)
    assert torch.equal(buf.flat_values_returns()[0], v_ref)
    # same permutation on every epoch, drawn from numpy's global RNG in reset()
    np.random.seed(4)
    assert np.array_equal(buf.indices, np.random.permutation(t * n))
