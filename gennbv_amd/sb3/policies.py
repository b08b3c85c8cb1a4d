"""ActorCriticPolicy_Train_Eval (stable_baselines3/common/policies.py:797-1090) for the
configuration GenNBV trains with: `net_arch=[]` (identity mlp_extractor), a features
extractor class + kwargs, MultiDiscrete actions, Adam(eps=1e-5) -- and, for BASELINE configs[0], SB3's
MLP policy (`FlattenExtractor` + `MlpExtractor`, sb3/torch_layers.py) on the plain-torch path.

Same constructor keywords, methods and `state_dict` keys
(`features_extractor.*`, `action_net.*`, `value_net.*`)."""
from __future__ import annotations

from functools import partial
from typing import Any, Dict, Optional, Tuple, Type

import numpy as np
import torch
from torch import nn

from .distributions import MultiCategoricalDistribution


class _IdentityExtractor(nn.Module):
    """MlpExtractor with net_arch=[] (torch_layers.py:217-228): no layers."""

    def __init__(self, feature_dim):
        super().__init__()
        self.latent_dim_pi = feature_dim
        self.latent_dim_vf = feature_dim

    def forward(self, features):
        return features, features

    def forward_actor(self, features):
        return features

    def forward_critic(self, features):
        return features


class ActorCriticPolicy_Train_Eval(nn.Module):
    def __init__(self, observation_space, action_space, lr_schedule, net_arch=None, activation_fn=nn.Tanh,
                 ortho_init: bool = True, use_sde: bool = False, features_extractor_class=None,
                 features_extractor_kwargs: Optional[Dict[str, Any]] = None, normalize_images: bool = True,
                 optimizer_class: Type[torch.optim.Optimizer] = torch.optim.Adam,
                 optimizer_kwargs: Optional[Dict[str, Any]] = None, **unused):
        super().__init__()
        assert not use_sde, "gSDE is not on the GenNBV path"
        # GenNBV trains with net_arch=[] (train_gennbv.py:150); a non-empty net_arch builds SB3's MlpExtractor (BASELINE configs[0]: the
        # CPU-runnable MLP policy, sb3/torch_layers.py).  net_arch=None is SB3's default for an MLP policy: two towers of 64, 64.
        # Defaults as the reference's (policies.py:844, :868-872): FlattenExtractor, two towers of 64, 64.
        if features_extractor_class is None:
            from .torch_layers import FlattenExtractor
            features_extractor_class = FlattenExtractor
        if net_arch is None:
            net_arch = [dict(pi=[64, 64], vf=[64, 64])]
        if optimizer_kwargs is None:
            optimizer_kwargs = {}
            if optimizer_class == torch.optim.Adam:
                optimizer_kwargs["eps"] = 1e-5  # policies.py:851-855
        self.observation_space, self.action_space = observation_space, action_space
        self.optimizer_class, self.optimizer_kwargs = optimizer_class, optimizer_kwargs
        self.ortho_init = ortho_init
        self.features_extractor = features_extractor_class(observation_space, **(features_extractor_kwargs or {}))
        self.features_dim = self.features_extractor.features_dim
        if len(net_arch) == 0:
            self.mlp_extractor = _IdentityExtractor(self.features_dim)
        else:
            from .torch_layers import MlpExtractor
            self.mlp_extractor = MlpExtractor(self.features_dim, net_arch, activation_fn)
        self.action_dist = MultiCategoricalDistribution(list(action_space.nvec))
        self.action_net = self.action_dist.proba_distribution_net(latent_dim=self.mlp_extractor.latent_dim_pi)
        self.value_net = nn.Linear(self.mlp_extractor.latent_dim_vf, 1)
        if ortho_init:
            # policies.py:983-994: Linear/Conv2d only -- Conv3d keeps torch's default init
            for module, gain in ((self.features_extractor, np.sqrt(2)), (self.mlp_extractor, np.sqrt(2)),
                                 (self.action_net, 0.01), (self.value_net, 1)):
                module.apply(partial(self.init_weights, gain=gain))
        self.optimizer = optimizer_class(self.parameters(), lr=lr_schedule(1), **optimizer_kwargs)

    @staticmethod
    def init_weights(module: nn.Module, gain: float = 1) -> None:
        if isinstance(module, (nn.Linear, nn.Conv2d)):
            nn.init.orthogonal_(module.weight, gain=gain)
            if module.bias is not None:
                module.bias.data.fill_(0.0)

    @property
    def device(self):
        return next(self.parameters()).device

    def set_training_mode(self, mode: bool) -> None:
        self.train(mode)

    def extract_features(self, obs: torch.Tensor) -> torch.Tensor:
        return self.features_extractor(obs.float())  # preprocess_obs: Box -> .float()

    def _dist(self, latent_pi):
        return self.action_dist.proba_distribution(action_logits=self.action_net(latent_pi))

    def _fused_head(self, obs):
        """(logits, values [B]) through the fused policy-head kernels, or None when not applicable
        (rollout on the gfx950 backend only: no autograd graph is needed there)."""
        if not getattr(self, "_fused_rollout", False) or torch.is_grad_enabled() or not obs.is_cuda:
            return None if isinstance(obs, torch.Tensor) else self._fused_head(obs.materialize())
        from ..ops import encoder_ops
        enc = self.features_extractor
        fa, fg = encoder_ops.hybrid_branches(enc, obs.float())
        logits, values, _ = encoder_ops.policy_head(enc, self.action_net, self.value_net, fa, fg)
        return logits, values

    def forward(self, obs: torch.Tensor, deterministic: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        fused = self._fused_head(obs)
        if fused is not None:
            logits, values = fused
            actions, log_prob = self.action_dist.sample_and_log_prob(logits, deterministic)
            return actions, values.unsqueeze(1), log_prob
        features = self.extract_features(obs)
        latent_pi, latent_vf = self.mlp_extractor(features)
        values = self.value_net(latent_vf)
        distribution = self._dist(latent_pi)
        actions = distribution.get_actions(deterministic=deterministic)
        return actions, values, distribution.log_prob(actions)

    def evaluate_actions(self, obs: torch.Tensor, actions: torch.Tensor):
        features = self.extract_features(obs)
        latent_pi, latent_vf = self.mlp_extractor(features)
        distribution = self._dist(latent_pi)
        log_prob = distribution.log_prob(actions)
        return self.value_net(latent_vf), log_prob, distribution.entropy()

    def get_distribution(self, obs: torch.Tensor):
        return self._dist(self.mlp_extractor.forward_actor(self.extract_features(obs)))

    def predict_values(self, obs: torch.Tensor) -> torch.Tensor:
        fused = self._fused_head(obs)
        if fused is not None:
            return fused[1].unsqueeze(1)
        return self.value_net(self.mlp_extractor.forward_critic(self.extract_features(obs)))
