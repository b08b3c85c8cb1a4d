"""evaluate_policy_grid_obs (stable_baselines3/common/evaluation.py:136-355) for tensor envs.

The reference hard-codes 50 eval envs and 30 steps per episode (:201-202) and reads a 5-tuple
(obs, rewards, dones, infos, accuracies) from an eval env whose step() returns the Chamfer accuracy of the
envs that just finished.  Here `n_envs` / `max_length` default to the env's own values, and the env may
return either the reference's 5-tuple or the training 4-tuple (accuracies then come from `accuracy_fn`)."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .metrics import auc_update, mean_auc


def evaluate_policy_grid_obs(model, env, n_eval_episodes: int = 10, deterministic: bool = True, return_AUC: bool = True,
                             max_length: Optional[int] = None, accuracy_fn: Optional[Callable] = None,
                             callback: Optional[Callable] = None):
    """-> (episode_rewards, episode_lengths, mean_AUC [n_envs] or None, episode_accuracies).
    `callback(locals, globals)` is called once per env step for every env still being counted, with the reference's
    local names `i`, `reward`, `done`, `info` (evaluation.py:291-307; `info` is the shared infos dict of a tensor env)."""
    n_envs = env.num_envs
    max_length = int(max_length if max_length is not None else env.max_episode_length)
    targets = torch.tensor([(n_eval_episodes + i) // n_envs for i in range(n_envs)], dtype=torch.int64)
    counts = torch.zeros(n_envs, dtype=torch.int64)
    if return_AUC:
        assert int(targets.max()) <= 1, "AUC evaluation runs at most one episode per env (evaluation.py:281)"
    cur_r, cur_l = torch.zeros(n_envs), torch.zeros(n_envs, dtype=torch.int64)
    out = env.reset()
    obs = out[0] if isinstance(out, tuple) else out
    auc = torch.zeros(n_envs, max_length)
    done_flag = torch.zeros(n_envs)
    ep_r, ep_l, ep_acc = [], [], []
    step = 0
    policy = model.policy
    while bool((counts < targets).any()):
        step += 1
        with torch.no_grad():
            actions, _, _ = policy(obs, deterministic=deterministic)  # == model.predict(obs, deterministic)
        out = env.step(actions)
        obs, rewards, dones = out[0], out[1].detach().float().cpu(), out[2].detach().cpu()
        infos = out[3] if len(out) > 3 else {}
        accuracies = out[4] if len(out) > 4 else None
        if return_AUC and step <= max_length:
            auc = auc_update(auc, rewards, step, dones, done_flag)
        cur_r += rewards
        cur_l += 1
        live = counts < targets
        if callback is not None:
            for i in torch.nonzero(live).flatten().tolist():
                info = infos if isinstance(infos, dict) else infos[i if n_envs == 1 else 0]
                callback(dict(i=i, reward=rewards[i], done=dones[i], info=info, infos=infos, rewards=rewards, dones=dones), globals())
        done_flag += (dones.bool() & live).float()
        for i in torch.nonzero(dones.bool() & live).flatten().tolist():
            ep_r.append(float(cur_r[i]))
            ep_l.append(int(cur_l[i]))
            if accuracies is not None:
                ep_acc.append(accuracies[str(i)] if isinstance(accuracies, dict) else float(accuracies[i]))
            elif accuracy_fn is not None:
                ep_acc.append(float(accuracy_fn(i)))
            counts[i] += 1
            cur_r[i] = 0
            cur_l[i] = 0
    return ep_r, ep_l, (mean_auc(auc) if return_AUC else None), ep_acc
