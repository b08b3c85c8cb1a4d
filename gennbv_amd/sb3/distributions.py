"""MultiCategoricalDistribution (stable_baselines3/common/distributions.py:299-352).

Same protocol (proba_distribution_net / proba_distribution / log_prob / entropy /
sample / mode / get_actions); the six categoricals are evaluated on the flat logits
[B, sum(action_dims)] in one pass instead of six torch.distributions.Categorical
objects."""
from __future__ import annotations

from typing import List

import torch
from torch import nn


class MultiCategoricalDistribution:
    def __init__(self, action_dims: List[int]):
        self.action_dims = [int(a) for a in action_dims]
        self._logp = None  # list of normalised log-probs per sub-space

    def proba_distribution_net(self, latent_dim: int) -> nn.Module:
        return nn.Linear(latent_dim, sum(self.action_dims))

    def proba_distribution(self, action_logits: torch.Tensor) -> "MultiCategoricalDistribution":
        # Categorical(logits=split) normalises: logits - logsumexp(logits)
        self._logp = [s - s.logsumexp(dim=-1, keepdim=True) for s in torch.split(action_logits, self.action_dims, dim=1)]
        return self

    def log_prob(self, actions: torch.Tensor) -> torch.Tensor:
        cols = torch.unbind(actions, dim=1)
        lp = [lg.gather(-1, a.long().unsqueeze(-1)).squeeze(-1) for lg, a in zip(self._logp, cols)]
        return torch.stack(lp, dim=1).sum(dim=1)

    def entropy(self) -> torch.Tensor:
        ent = []
        for lg in self._logp:
            lg = torch.clamp(lg, min=torch.finfo(lg.dtype).min)
            ent.append(-(lg * lg.exp()).sum(-1))
        return torch.stack(ent, dim=1).sum(dim=1)

    def sample(self) -> torch.Tensor:
        # torch.distributions.Categorical.sample == multinomial(probs, 1): same RNG consumption order
        return torch.stack([torch.multinomial(lg.exp(), 1, True).squeeze(-1) for lg in self._logp], dim=1)

    def mode(self) -> torch.Tensor:
        return torch.stack([torch.argmax(lg.exp(), dim=1) for lg in self._logp], dim=1)

    def get_actions(self, deterministic: bool = False) -> torch.Tensor:
        return self.mode() if deterministic else self.sample()

    def sample_and_log_prob(self, action_logits: torch.Tensor, deterministic: bool = False):
        """(actions [B, H] int64, log_prob [B]) of the rollout forward in ONE launch on the GPU
        (gnbv_multicategorical_sample): inverse CDF of softmax at one uniform per (row, head).  Same
        distribution as `sample()` (torch.multinomial), a different random stream -- the CPU path above keeps
        the reference's RNG consumption order."""
        import ctypes as C

        from .. import _lib
        lib = _lib.load()
        _lib.require_cuda(action_logits)
        logits = action_logits.contiguous().float()
        b, h = logits.shape[0], len(self.action_dims)
        dev = logits.device
        u = None if deterministic else torch.rand(b, h, device=dev)
        actions = torch.empty(b, h, dtype=torch.int64, device=dev)
        log_prob = torch.empty(b, dtype=torch.float32, device=dev)
        dims = (C.c_int * h)(*self.action_dims)
        _lib.check(lib.gnbv_multicategorical_sample(logits.data_ptr(), b, logits.shape[1], h, dims, _lib.ptr(u), int(deterministic),
                                                    actions.data_ptr(), log_prob.data_ptr(), _lib.stream_ptr(dev)),
                   "gnbv_multicategorical_sample")
        return actions, log_prob
