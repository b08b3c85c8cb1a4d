"""Helpers shared by the policy / PPO parity tests."""
import numpy as np
import torch

from tests import golden_util as gu
from gennbv_amd.spaces import Box, MultiDiscrete

G, STACK = 20, 100
NVEC = [81, 81, 51, 1, 13, 13]


def obs_dim(g=G):
    return STACK * 6 + g ** 3 + 8192


def spaces(g=G):
    return Box(-np.inf, np.inf, shape=(obs_dim(g),), dtype=np.float32), MultiDiscrete(NVEC)


def policy_kwargs(g=G, backend="torch", **enc_kw):
    from tests.torch_reference import encoder_class
    return dict(net_arch=[], features_extractor_class=encoder_class(backend),
                features_extractor_kwargs=dict(encoder_param={"hidden_shapes": [256, 256], "visual_dim": 256},
                                               net_param={"transformer_params": [[1, 256], [1, 256]],
                                                          "append_hidden_shapes": [256, 256]},
                                               state_input_shape=(STACK * 6,), visual_input_shape=(STACK, 400, 400),
                                               grid_size=g, **enc_kw))


def make_policy(g=G, device="cpu", backend="torch", det_weights=True, **enc_kw):
    from gennbv_amd.sb3.policies import ActorCriticPolicy_Train_Eval
    obs_space, act_space = spaces(g)
    kw = policy_kwargs(g, backend, **enc_kw)
    pol = ActorCriticPolicy_Train_Eval(obs_space, act_space, lambda _: 1e-4, **kw)
    if det_weights:
        shapes = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
        pol.load_state_dict({k: torch.from_numpy(v) for k, v in gu.det_state_dict(shapes).items()})
    return pol.to(device), obs_space, act_space


def unpack_obs(fx):
    return torch.from_numpy(np.concatenate([fx["obs_state"], fx["obs_grid"].astype(np.float32),
                                            fx["obs_rgb"].astype(np.float32)], axis=1))
