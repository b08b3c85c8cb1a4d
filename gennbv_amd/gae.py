"""GAE scans on the gfx950 kernel (gennbv_amd/csrc/gae.hip).

compute_returns_and_advantage : stable_baselines3/common/buffers.py:706-724
compute_returns_rsl           : rsl_rl/storage/rollout_storage.py:130-144
"""
from __future__ import annotations

import torch

from . import _lib


def compute_returns_and_advantage(rewards, values, episode_starts, last_values, dones, gamma: float, gae_lambda: float,
                                  advantages=None, returns=None):
    """All inputs [T,N] or [T,N,1] (episode_starts uint8/bool, dones [N]); returns
    (advantages, returns) shaped like `rewards`."""
    _lib.require_cuda(rewards, values, episode_starts, last_values, dones)
    lib = _lib.load()
    t, n = rewards.shape[0], rewards.shape[1]
    r, v = rewards.contiguous().float(), values.contiguous().float()
    es = episode_starts.contiguous().to(torch.uint8)
    lv = last_values.detach().contiguous().float().view(-1)
    dn = dones.contiguous().to(torch.uint8).view(-1)
    assert lv.numel() == n and dn.numel() == n
    advantages = torch.empty_like(r) if advantages is None else advantages
    returns = torch.empty_like(r) if returns is None else returns
    _lib.check(lib.gnbv_gae_sb3(r.data_ptr(), v.data_ptr(), es.data_ptr(), lv.data_ptr(), dn.data_ptr(), t, n,
                                float(gamma), float(gae_lambda), advantages.data_ptr(), returns.data_ptr(),
                                _lib.stream_ptr(r.device)), "gnbv_gae_sb3")
    return advantages, returns


def compute_returns_rsl(rewards, values, dones, last_values, gamma: float, lam: float, normalize: bool = True):
    """rsl_rl convention; returns (returns, advantages) with the whole-buffer
    normalisation of rollout_storage.py:143-144 applied when `normalize`."""
    _lib.require_cuda(rewards, values, dones, last_values)
    lib = _lib.load()
    t, n = rewards.shape[0], rewards.shape[1]
    r, v = rewards.contiguous().float(), values.contiguous().float()
    dn = dones.contiguous().to(torch.uint8)
    lv = last_values.detach().contiguous().float().view(-1)
    ret, adv = torch.empty_like(r), torch.empty_like(r)
    _lib.check(lib.gnbv_gae_rsl(r.data_ptr(), v.data_ptr(), dn.data_ptr(), lv.data_ptr(), t, n, float(gamma), float(lam),
                                ret.data_ptr(), adv.data_ptr(), _lib.stream_ptr(r.device)), "gnbv_gae_rsl")
    if normalize:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    return ret, adv
