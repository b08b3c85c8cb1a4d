"""Generate the encoder / policy / PPO goldens by RUNNING THE REFERENCE (this container only).

TEST INFRASTRUCTURE ONLY.   python oracle/gen_golden_ppo.py

  F7_policy     B1/B2  gennbv/network/hybrid_encoder.py:18-98 +
                       stable_baselines3/common/policies.py:797-1090 (ActorCriticPolicy_Train_Eval,
                       MultiCategoricalDistribution): features / values / logits / log_prob / entropy in
                       eval and train (BatchNorm batch-stat) mode, gradients of a scalar loss
  F9_ppo_train  C1/C3/C4  TensorRolloutBuffer_Grid_Obs.get + PPO_Grid_Obs.train()
                       (stable_baselines3/ppo/ppo_grid_obs.py:176-297) on a recorded rollout buffer:
                       logged losses, parameter trajectory, BN running stats, early-stop position
The reference policy only exists at G=20 (hard-coded 8000 / 20 / 1024, hybrid_encoder.py:47,90-91).
"""
from __future__ import annotations

import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
from tests import golden_util as gu  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
G, STACK = 20, 100
D_OBS = STACK * 6 + G ** 3 + 8192
NVEC = [81, 81, 51, 1, 13, 13]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_policy(ref, seed=0):
    import gym
    torch.manual_seed(seed)
    obs_space = gym.spaces.Box(-np.inf, np.inf, shape=(D_OBS,), dtype=np.float32)
    act_space = gym.spaces.MultiDiscrete(NVEC)
    kw = dict(net_arch=[], features_extractor_class=ref.hybrid_encoder.Hybrid_Encoder,
              features_extractor_kwargs=dict(encoder_param={"hidden_shapes": [256, 256], "visual_dim": 256},
                                             net_param={"transformer_params": [[1, 256], [1, 256]],
                                                        "append_hidden_shapes": [256, 256]},
                                             state_input_shape=(STACK * 6,), visual_input_shape=(STACK, 400, 400)))
    pol = ref.policies.ActorCriticPolicy_Train_Eval(obs_space, act_space, lambda _: 1e-4, **kw)
    # deterministic platform-independent weights (tests/golden_util.det_state_dict): nothing to store
    shapes = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in gu.det_state_dict(shapes).items()})
    return pol, obs_space, act_space


def make_obs(b, gen):
    """Observation rows with realistic content: lattice poses, ternary grid, gray frames."""
    unit = torch.tensor([0.2, 0.2, 0.2, 0.0, 3.14159265359 / 12, 3.14159265359 / 6])
    low = torch.tensor([-8.0, -8.0, 0.1, 0.0, -3.14159265359 / 2, 0.0])
    a = torch.stack([torch.randint(0, n, (b, STACK), generator=gen) for n in NVEC], -1).float()
    state = (a * unit + low).reshape(b, -1)
    grid = torch.randint(-1, 2, (b, G ** 3), generator=gen).float()
    grid = grid * (torch.rand(b, G ** 3, generator=gen) < 0.4).float()
    rgb = torch.randint(0, 256, (b, 8192), generator=gen).float()
    return torch.cat([state, grid, rgb], 1), a


def pack_obs(obs):
    st = obs[:, :600].numpy().astype(np.float32)
    gr = obs[:, 600:600 + G ** 3].numpy().astype(np.int8)
    rg = obs[:, 600 + G ** 3:].numpy().astype(np.uint8)
    return st, gr, rg


def gen_policy(ref):
    pol, _, _ = make_policy(ref, seed=0)
    gen = torch.Generator().manual_seed(1)
    b = 8
    obs, _ = make_obs(b, gen)
    actions = torch.stack([torch.randint(0, n, (b,), generator=gen) for n in NVEC], -1).float()
    out = {}
    sd0 = {k: v.clone() for k, v in pol.state_dict().items()}
    out["sd_names"] = np.array(list(sd0.keys()))
    out["sd_sha"] = np.array([sha(v.numpy()) for v in sd0.values()])
    st, gr, rg = pack_obs(obs)
    out.update(obs_state=st, obs_grid=gr, obs_rgb=rg, actions=actions.numpy())
    # eval mode (rollout): BatchNorm running stats
    pol.set_training_mode(False)
    with torch.no_grad():
        feats = pol.extract_features(obs)
        values, log_prob, entropy = pol.evaluate_actions(obs, actions)
        logits = pol.action_net(feats)
        pv = pol.predict_values(obs)
    out.update(eval_features=feats.numpy(), eval_values=values.numpy(), eval_logits=logits.numpy(),
               eval_log_prob=log_prob.numpy(), eval_entropy=entropy.numpy(), eval_predict_values=pv.numpy())
    # train mode: batch statistics, gradients of a scalar loss, running-stat update
    pol.set_training_mode(True)
    pol.zero_grad()
    values, log_prob, entropy = pol.evaluate_actions(obs, actions)
    w = torch.linspace(0.5, 1.5, b)
    loss = (values.flatten() * w).sum() + (log_prob * w.flip(0)).sum() + 0.3 * (entropy * w).sum()
    loss.backward()
    out.update(train_values=values.detach().numpy(), train_log_prob=log_prob.detach().numpy(),
               train_entropy=entropy.detach().numpy(), train_loss=np.float64(loss.item()))
    for pname, p in pol.named_parameters():
        gr = p.grad.numpy()
        out["grad_norm/" + pname] = np.float64(np.sqrt((gr.astype(np.float64) ** 2).sum()))
        out["grad/" + pname] = gr.copy() if gr.size <= 70000 else gr.reshape(-1)[::97].copy()
    for k, v in pol.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["bn_after/" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(GOLDEN, "F7_policy.npz"), **out)
    print("F7_policy saved; n_params", sum(p.numel() for p in pol.parameters()))
    return sd0


def gen_ppo_train(ref, sd0, name, target_kl, n_epochs, lr):
    pol, obs_space, act_space = make_policy(ref, seed=0)
    pol.load_state_dict(sd0)
    pol.optimizer = torch.optim.Adam(pol.parameters(), lr=lr, eps=1e-5)
    T, N, BS = 8, 4, 8
    gen = torch.Generator().manual_seed(7)
    Buf = ref.buffers.TensorRolloutBuffer_Grid_Obs
    np.random.seed(123)
    buf = Buf(T, obs_space, act_space, device="cpu", gamma=0.99, gae_lambda=0.95, n_envs=N)
    indices = buf.indices.copy()
    obs, _ = make_obs(T * N, gen)
    obs = obs.view(T, N, -1)
    actions = torch.stack([torch.randint(0, n, (T, N), generator=gen) for n in NVEC], -1).float()
    pol.set_training_mode(False)
    with torch.no_grad():
        values, log_probs, _ = pol.evaluate_actions(obs.view(T * N, -1), actions.view(T * N, -1))
    values = values.view(T, N, 1) + 0.05 * torch.randn(T, N, 1, generator=gen)
    log_probs = log_probs.view(T, N) + 0.02 * torch.randn(T, N, generator=gen)
    rewards = torch.rand(T, N, generator=gen) * 0.5
    starts = torch.rand(T, N, generator=gen) < 0.1
    starts[0] = True
    for t in range(T):
        buf.add(obs[t], actions[t], rewards[t], starts[t].numpy(), values[t], log_probs[t])
    last_values = torch.randn(N, 1, generator=gen) * 0.1
    dones = torch.zeros(N, dtype=torch.long)
    buf.compute_returns_and_advantage(last_values=last_values, dones=dones)
    adv, ret = buf.advantages.clone(), buf.returns.clone()

    PPO = ref.ppo_grid_obs.PPO_Grid_Obs
    ppo = object.__new__(PPO)
    ppo.policy, ppo.rollout_buffer = pol, buf
    ppo.batch_size, ppo.n_epochs = BS, n_epochs
    ppo.clip_range = lambda _: 0.2
    ppo.clip_range_vf = lambda _: 0.2
    ppo.normalize_advantage, ppo.ent_coef, ppo.vf_coef = True, 0.01, 0.8
    ppo.max_grad_norm, ppo.target_kl = 1.0, target_kl
    ppo.action_space, ppo.use_sde = act_space, False
    ppo._current_progress_remaining, ppo._n_updates, ppo.verbose = 1.0, 0, 0
    ppo.lr_schedule = lambda _: lr
    rec = {}
    ppo._logger = types.SimpleNamespace(record=lambda k, v, exclude=None: rec.__setitem__(k, v))
    ppo._custom_logger = True
    traj = []
    orig_step = pol.optimizer.step

    def step_hook(*a, **k):
        r = orig_step(*a, **k)
        traj.append([float(p.detach().double().sum()) for p in pol.parameters()])
        return r
    pol.optimizer.step = step_hook
    PPO.train(ppo)
    out = {}
    for k, v in sd0.items():
        pass  # initial weights live in F7_policy.npz (same seed)
    st, gr, rg = pack_obs(obs.view(T * N, -1))
    out.update(obs_state=st, obs_grid=gr, obs_rgb=rg, actions=actions.numpy(), values=values.numpy(),
               log_probs=log_probs.numpy(), rewards=rewards.numpy(), episode_starts=starts.numpy().astype(np.uint8),
               last_values=last_values.numpy(), dones=dones.numpy().astype(np.uint8), indices=indices,
               advantages=adv.numpy().reshape(T, N), returns=ret.numpy().reshape(T, N),
               T=T, N=N, batch_size=BS, n_epochs=n_epochs, lr=lr, target_kl=-1.0 if target_kl is None else target_kl,
               n_optimizer_steps=len(traj), param_sum_trajectory=np.array(traj, np.float64))
    for k, v in rec.items():
        if isinstance(v, (int, float, np.floating, np.integer)):
            out["log/" + k] = np.float64(v)
    for pname, p in pol.named_parameters():
        a = p.detach().numpy()
        out["final_sha/" + pname] = sha(a)
        out["final/" + pname] = a if a.size <= 70000 else a.reshape(-1)[::97].copy()
    for k, v in pol.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["final_bn/" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    print(name, "saved; optimizer steps", len(traj), {k: rec[k] for k in rec if k.startswith("train/")})


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    ref = ref_harness.import_reference()
    torch.set_num_threads(8)
    sd0 = gen_policy(ref)
    gen_ppo_train(ref, sd0, "F9_ppo_train", target_kl=None, n_epochs=3, lr=1e-4)
    gen_ppo_train(ref, sd0, "F9_ppo_train_earlystop", target_kl=0.05, n_epochs=3, lr=3e-4)
