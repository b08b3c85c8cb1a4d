"""Generate the MLP-policy golden of BASELINE configs[0] by RUNNING THE REFERENCE (this container only).

TEST INFRASTRUCTURE ONLY.   python oracle/gen_golden_mlp.py

  F15_ppo_mlp_c0   configs[0]: 4 envs, 16^3 grid, MLP policy, PPO on the CPU.  The reference's own ActorCriticPolicy_Train_Eval
                   with its defaults for an MLP policy (stable_baselines3/common/policies.py:844 FlattenExtractor, :868-872
                   net_arch = [dict(pi=[64, 64], vf=[64, 64])], torch_layers.py:135-240 MlpExtractor) and a second architecture with a
                   shared layer ([96, dict(pi=[48], vf=[32, 16])]), evaluated in eval mode (values / logits / log-prob / entropy)
                   and trained by the reference's own PPO_Grid_Obs.train() (ppo_grid_obs.py:176-297) on a recorded 8 x 4 rollout
                   buffer of 16^3 observation rows: logged losses, parameter trajectory, final parameters.
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
G, STACK = 16, 100
D_OBS = STACK * 6 + G ** 3 + 8192
NVEC = [81, 81, 51, 1, 13, 13]
ARCHS = {"default": None, "shared": [96, dict(pi=[48], vf=[32, 16])]}


def make_policy(ref, net_arch):
    import gym
    obs_space = gym.spaces.Box(-np.inf, np.inf, shape=(D_OBS,), dtype=np.float32)
    act_space = gym.spaces.MultiDiscrete(NVEC)
    kw = {} if net_arch is None else {"net_arch": net_arch}
    pol = ref.policies.ActorCriticPolicy_Train_Eval(obs_space, act_space, lambda _: 1e-5, **kw)
    shapes = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in gu.det_state_dict(shapes).items()})
    return pol, obs_space, act_space


def make_obs(b, gen):
    unit = torch.tensor([0.2, 0.2, 0.2, 0.0, 3.14159265359 / 12, 3.14159265359 / 6])
    low = torch.tensor([-8.0, -8.0, 0.1, 0.0, -3.14159265359 / 2, 0.0])
    a = torch.stack([torch.randint(0, n, (b, STACK), generator=gen) for n in NVEC], -1).float()
    state = (a * unit + low).reshape(b, -1)
    grid = torch.randint(-1, 2, (b, G ** 3), generator=gen).float() * (torch.rand(b, G ** 3, generator=gen) < 0.4).float()
    rgb = torch.randint(0, 256, (b, 8192), generator=gen).float() / 255.0  # (an MLP on raw 0..255 gray values saturates its tanh layers)
    return torch.cat([state, grid, rgb], 1)


def run(ref, tag, net_arch, out):
    pol, obs_space, act_space = make_policy(ref, net_arch)
    lr, T, N, BS, n_epochs = 1e-5, 8, 4, 8, 3
    pol.optimizer = torch.optim.Adam(pol.parameters(), lr=lr, eps=1e-5)
    out[tag + "/sd_names"] = np.array(list(pol.state_dict().keys()))
    out[tag + "/sd_shapes"] = np.array([str(tuple(v.shape)) for v in pol.state_dict().values()])
    gen = torch.Generator().manual_seed(11)
    obs = make_obs(T * N, gen).view(T, N, -1)
    actions = torch.stack([torch.randint(0, n, (T, N), generator=gen) for n in NVEC], -1).float()
    pol.set_training_mode(False)
    with torch.no_grad():
        values, log_probs, entropy = pol.evaluate_actions(obs.view(T * N, -1), actions.view(T * N, -1))
        logits = pol.action_net(pol.mlp_extractor.forward_actor(pol.extract_features(obs.view(T * N, -1))))
    out.update({tag + "/eval_values": values.numpy(), tag + "/eval_log_prob": log_probs.numpy(), tag + "/eval_entropy": entropy.numpy(),
                tag + "/eval_logits": logits.numpy()})
    Buf = ref.buffers.TensorRolloutBuffer_Grid_Obs
    np.random.seed(321)
    buf = Buf(T, obs_space, act_space, device="cpu", gamma=0.99, gae_lambda=0.95, n_envs=N)
    indices = buf.indices.copy()
    v_buf = values.view(T, N, 1) + 0.05 * torch.randn(T, N, 1, generator=gen)
    lp_buf = log_probs.view(T, N) + 0.02 * torch.randn(T, N, generator=gen)
    rewards = torch.rand(T, N, generator=gen) * 0.5
    starts = torch.rand(T, N, generator=gen) < 0.1
    starts[0] = True
    for t in range(T):
        buf.add(obs[t], actions[t], rewards[t], starts[t].numpy(), v_buf[t], lp_buf[t])
    last_values = torch.randn(N, 1, generator=gen) * 0.1
    dones = torch.zeros(N, dtype=torch.long)
    buf.compute_returns_and_advantage(last_values=last_values, dones=dones)
    adv, ret = buf.advantages.clone(), buf.returns.clone()  # (before train(): the reference's get() swaps and flattens its arrays in place)
    PPO = ref.ppo_grid_obs.PPO_Grid_Obs
    ppo = object.__new__(PPO)
    ppo.policy, ppo.rollout_buffer = pol, buf
    ppo.batch_size, ppo.n_epochs = BS, n_epochs
    ppo.clip_range = lambda _: 0.2
    ppo.clip_range_vf = lambda _: 0.2
    ppo.normalize_advantage, ppo.ent_coef, ppo.vf_coef = True, 0.01, 0.8
    ppo.max_grad_norm, ppo.target_kl = 1.0, None
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
    if tag == "default":  # (the buffer is the same for both architectures up to values / log-probs)
        flat = obs.view(T * N, -1)
        out.update(obs_state=flat[:, :600].numpy().astype(np.float32), obs_grid=flat[:, 600:600 + G ** 3].numpy().astype(np.int8),
                   obs_rgb_u8=(flat[:, 600 + G ** 3:] * 255.0).round().numpy().astype(np.uint8), actions=actions.numpy(), rewards=rewards.numpy(),
                   episode_starts=starts.numpy().astype(np.uint8), last_values=last_values.numpy(), dones=dones.numpy().astype(np.uint8),
                   indices=indices, T=T, N=N, batch_size=BS, n_epochs=n_epochs, lr=lr, G=G)
    out.update({tag + "/values": v_buf.numpy(), tag + "/log_probs": lp_buf.numpy(), tag + "/advantages": adv.numpy().reshape(T, N),
                tag + "/returns": ret.numpy().reshape(T, N), tag + "/n_optimizer_steps": len(traj),
                tag + "/param_sum_trajectory": np.array(traj, np.float64)})
    for k, v in rec.items():
        if isinstance(v, (int, float, np.floating, np.integer)):
            out[tag + "/log/" + k] = np.float64(v)
    for pname, p in pol.named_parameters():
        a = p.detach().numpy()
        out[tag + "/final/" + pname] = a if a.size <= 70000 else a.reshape(-1)[::97].copy()
    print(tag, "optimizer steps", len(traj), {k: round(float(rec[k]), 6) for k in rec if k.startswith("train/")})


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    ref = ref_harness.import_reference()
    # (the harness stubs gym; FlattenExtractor asks gym.spaces.utils.flatdim for the row length: a Box's is the product of its shape)
    sys.modules["stable_baselines3.common.torch_layers"].get_flattened_obs_dim = lambda sp: int(np.prod(sp.shape))
    torch.set_num_threads(8)
    out = {}
    for tag, arch in ARCHS.items():
        run(ref, tag, arch, out)
    np.savez_compressed(os.path.join(GOLDEN, "F15_ppo_mlp_c0.npz"), **out)
    print("F15_ppo_mlp_c0 saved,", os.path.getsize(os.path.join(GOLDEN, "F15_ppo_mlp_c0.npz")) // 1024, "KiB")
