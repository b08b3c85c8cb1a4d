"""The two small SB3 building blocks an MLP policy needs (stable_baselines3/common/torch_layers.py:34-46 `FlattenExtractor`, :135-240
`MlpExtractor`), for BASELINE configs[0]: 4 envs, 16^3 grid, MLP policy, PPO on the CPU over a recorded feed.  Plain torch modules -- this
configuration never touches the gfx950 kernels (the fused rollout / train paths of PPO_Grid_Obs require the identity extractor and the
Hybrid_Encoder) -- with the reference's module tree, i.e. its `state_dict` keys: `mlp_extractor.shared_net.N.*`, `.policy_net.N.*`,
`.value_net.N.*`."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Type, Union

import numpy as np
import torch
from torch import nn


class FlattenExtractor(nn.Module):
    """features = the observation row itself (features_dim = its length)."""

    def __init__(self, observation_space):
        super().__init__()
        self.features_dim = int(np.prod(observation_space.shape))
        self.flatten = nn.Flatten()

    def forward(self, observations: torch.Tensor) -> torch.Tensor:
        return self.flatten(observations)


def _tower(widths: Sequence[int], d_in: int, act: Type[nn.Module]) -> Tuple[nn.Sequential, int]:
    mods: List[nn.Module] = []
    for w in widths:
        if not isinstance(w, int):
            raise TypeError("layer widths must be integers")
        mods += [nn.Linear(d_in, w), act()]
        d_in = w
    return nn.Sequential(*mods), d_in


class MlpExtractor(nn.Module):
    """net_arch = [shared widths ..., {"pi": [...], "vf": [...]}]: integers in front are layers both heads share; the optional dict
    (everything behind it is ignored, as in the reference) lists the layers of the policy / value towers; a missing key = no layers.
    An empty tower is the identity, so `[64, 64]` gives latent_pi == latent_vf."""

    def __init__(self, feature_dim: int, net_arch: List[Union[int, Dict[str, List[int]]]], activation_fn: Type[nn.Module], device="auto"):
        super().__init__()
        shared: List[int] = []
        towers: Dict[str, List[int]] = {}
        for item in net_arch:
            if isinstance(item, int):
                shared.append(item)
                continue
            if not isinstance(item, dict):
                raise TypeError("net_arch holds integers and at most one dict(pi=..., vf=...)")
            for key in ("pi", "vf"):
                if key in item and not isinstance(item[key], list):
                    raise TypeError(f"net_arch[...]['{key}'] must be a list of integers")
            towers = item
            break
        self.shared_net, d = _tower(shared, feature_dim, activation_fn)
        # The two towers' layers are CREATED depth by depth, policy layer i then value layer i (the reference walks zip_longest over the
        # two width lists, stable_baselines3/common/torch_layers.py MlpExtractor.__init__): the parameters' default initialisation draws
        # from torch's global generator in creation order, so a seeded construction only matches the reference's for ortho_init=False
        # when the order does.  Module tree / state_dict keys are the same either way.
        pi_w, vf_w = list(towers.get("pi", [])), list(towers.get("vf", []))
        pi_mods, vf_mods, d_pi, d_vf = [], [], d, d
        for i in range(max(len(pi_w), len(vf_w))):
            for widths, mods, which in ((pi_w, pi_mods, "pi"), (vf_w, vf_mods, "vf")):
                if i < len(widths):
                    if not isinstance(widths[i], int):
                        raise TypeError("layer widths must be integers")
                    d_in = d_pi if which == "pi" else d_vf
                    mods += [nn.Linear(d_in, widths[i]), activation_fn()]
                    if which == "pi":
                        d_pi = widths[i]
                    else:
                        d_vf = widths[i]
        self.policy_net, self.latent_dim_pi = nn.Sequential(*pi_mods), d_pi
        self.value_net, self.latent_dim_vf = nn.Sequential(*vf_mods), d_vf
        if device not in ("auto", None):
            self.to(device)

    def forward(self, features: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        h = self.shared_net(features)
        return self.policy_net(h), self.value_net(h)

    def forward_actor(self, features: torch.Tensor) -> torch.Tensor:
        return self.policy_net(self.shared_net(features))

    def forward_critic(self, features: torch.Tensor) -> torch.Tensor:
        return self.value_net(self.shared_net(features))
