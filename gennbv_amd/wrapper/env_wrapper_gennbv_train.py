"""EnvWrapperGenNBVTrain counterpart (gennbv/wrapper/env_wrapper_gennbv_train.py:90-133).

The reference wrapper flattens the dict observation {state, grid, state_rgb} into one
row per env.  ReplayFeedEnv's kernels already write that flat layout in place, so this
class only carries the reference's name / protocol (`_gym_env`, attribute passthrough,
`reset()`, `step()`); `flatten_observations` is provided for dict-producing envs."""
from __future__ import annotations

import torch

KEY_SEQUENCE = ["state", "grid", "state_rgb"]


def flatten_observations(observation_dict, key_sequence=KEY_SEQUENCE):
    observations = []
    for key in key_sequence:
        value = observation_dict[key]
        assert key in ["state", "state_rgb", "grid"]
        observations.append(value.reshape(value.shape[0], -1))
    return torch.concat(observations, dim=-1)


class EnvWrapperGenNBVTrain:
    def __init__(self, gym_env, observation_excluded=()):
        self.observation_excluded = observation_excluded
        self._gym_env = gym_env
        self.observation_space = gym_env.observation_space
        self.action_space = gym_env.action_space

    def __getattr__(self, attr):
        return getattr(self._gym_env, attr)

    def __setattr__(self, k, v):
        # the algorithm assigns env.episode_length_buf (base_class_grid_obs.py:471-475)
        if k == "episode_length_buf":
            self._gym_env.episode_length_buf.copy_(v)
        else:
            object.__setattr__(self, k, v)

    def reset(self, **kw):
        return self._gym_env.reset(**kw)

    def step(self, action, **kw):
        return self._gym_env.step(action, **kw)

    def close(self):
        self._gym_env.close()
