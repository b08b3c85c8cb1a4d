"""Constants of the GenNBV training task (values, not code, from the reference config).

Reference: gennbv/env/config_gennbv_train.py:6-73 (Config_GenNBV_Train),
gennbv/train/train_gennbv.py:19-93,117-187 (CLI defaults = PPO hyper-parameters),
legged_gym/env/base/drone_robot.py:660-691,875 (reward scales are multiplied by
dt = decimation * sim_dt = 4 * 0.005).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List

PI = 3.14159265359  # the reference's own truncated constant (config_gennbv_train.py:59)


@dataclass
class TaskConfig:
    # visual input (config_gennbv_train.py:22-35; BASELINE configs use 240x320)
    camera_width: int = 400
    camera_height: int = 400
    horizontal_fov: float = 90.0
    stack: int = 100  # pose history depth (buffer_size)
    # occupancy grid edge (the reference loads it from the GT file: env_train_gennbv.py:82)
    grid_size: int = 20
    # grayscale frames kept in the observation (env_train_gennbv.py:192-197)
    rgb_k: int = 2
    rgb_h: int = 64
    rgb_w: int = 64
    # action lattice (config_gennbv_train.py:62-69)
    clip_pose_low: List[float] = field(default_factory=lambda: [-8.0, -8.0, 0.1, 0.0, -0.5 * PI, 0.0])
    clip_pose_idx_up: List[int] = field(default_factory=lambda: [80, 80, 50, 0, 12, 12])
    clip_pose_idx_low: List[int] = field(default_factory=lambda: [0, 0, 0, 0, 0, 0])
    init_pose_buf: List[float] = field(default_factory=lambda: [0.0, 0.0, 10.1, 0.0, 90.0 / 180.0 * math.pi, 0.0])
    init_action: List[int] = field(default_factory=lambda: [40, 40, 50, 0, 12, 0])
    action_unit: List[float] = field(default_factory=lambda: [0.2, 0.2, 0.2, 0.0, PI / 12.0, PI / 6.0])
    # episode / reward (config_gennbv_train.py:11-20; drone_robot.py:875)
    max_episode_length: int = 100
    episode_length_s: float = 20.0
    dt: float = 4 * 0.005
    scale_surface_coverage: float = 1000.0
    scale_short_path: float = 5.0
    scale_termination: float = 50.0
    only_positive_rewards: bool = True
    coverage_threshold: float = 0.99  # env_train_gennbv.py:455
    depth_sense_dist: float = -50.0  # env_train_base.py:24
    seg_fg_threshold: float = 50.0  # env_train_gennbv.py:504
    env_spacing: float = 5.0

    @property
    def action_nvec(self) -> List[int]:
        return [u - l + 1 for u, l in zip(self.clip_pose_idx_up, self.clip_pose_idx_low)]

    @property
    def state_dim(self) -> int:
        return self.stack * 6

    @property
    def grid_dim(self) -> int:
        return self.grid_size ** 3

    @property
    def rgb_dim(self) -> int:
        return self.rgb_k * self.rgb_h * self.rgb_w

    @property
    def obs_dim(self) -> int:
        """Flat observation width, key order state, grid, state_rgb
        (gennbv/wrapper/env_wrapper_gennbv_train.py:104-110)."""
        return self.state_dim + self.grid_dim + self.rgb_dim


@dataclass
class PPOConfig:
    """gennbv/train/train_gennbv.py:19-93,170-187 defaults."""
    learning_rate: float = 1e-4
    n_steps: int = 128
    batch_size: int = 128
    n_epochs: int = 5
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    clip_range_vf: float = 0.2
    ent_coef: float = 0.01
    vf_coef: float = 0.8
    max_grad_norm: float = 1.0
    target_kl: float = 0.05
    policy_loss_scale: float = 10.0  # stable_baselines3/ppo/ppo_grid_obs.py:253
    adam_eps: float = 1e-5  # stable_baselines3/common/policies.py:851-855


def baseline_config(idx: int) -> TaskConfig:
    """BASELINE.json `configs[idx]` (240x320 depth; grid 16/64/64/64/128)."""
    grid = {0: 16, 1: 64, 2: 64, 3: 64, 4: 128}[idx]
    return TaskConfig(camera_width=320, camera_height=240, grid_size=grid)
