"""CPU oracle of one Env_Train_GenNBV step on a recorded feed (numpy + oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c).  Restates, for a simulator-free
feed, exactly what the reference does between `env.step(actions)` and the
returned `(obs, rewards, dones, infos)`:

  step                    gennbv/env/env_train_gennbv.py:246-264
  post_physics_step       :328-344        (episode_length_buf += 1)
  get_step_return         :346-375
  post_process_camera...  gennbv/env/env_train_base.py:513-534
  update_obs_buf / update_occ_grid  env_train_gennbv.py:273-326
  compute_reward          env_train_base.py:377-398 ; _reward_* env_train_gennbv.py:535-556
  check_termination       env_train_gennbv.py:438-457
  reset_idx               :377-436
  flatten_observations    gennbv/wrapper/env_wrapper_gennbv_train.py:27-56,104-110

Pinned by tests/golden/F5_envstep_*.npz (the reference env run on a fake simulator).
Reference quirk kept on purpose: `infos["time_outs"]` is only refreshed on steps in
which at least one env resets (reset_idx returns early on an empty id list, :390-391,
so `extras["time_outs"]` keeps pointing at an older tensor otherwise).
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc

f32 = np.float32


class OracleEnv:
    def __init__(self, cfg, inv_intri, range_gt, voxel_size, grid_gt, num_valid_voxel_gt, max_episode_length=None):
        self.cfg = cfg
        self.n = grid_gt.shape[0]
        self.g = cfg.grid_size
        self.inv_intri = np.ascontiguousarray(inv_intri, f32)
        self.range_gt = np.ascontiguousarray(range_gt, f32)
        self.voxel_size = np.ascontiguousarray(voxel_size, f32)
        self.grid_gt = np.ascontiguousarray(grid_gt, f32)
        self.num_valid = np.ascontiguousarray(num_valid_voxel_gt, f32)
        self.max_episode_length = cfg.max_episode_length if max_episode_length is None else max_episode_length
        n, g = self.n, self.g
        self.action_unit = np.array(cfg.action_unit, f32)
        self.action_low = np.array(cfg.clip_pose_low, f32)
        self.clip_low = np.array(cfg.clip_pose_idx_low, np.int64)
        self.clip_up = np.array(cfg.clip_pose_idx_up, np.int64)
        self.init_action = np.array(cfg.init_action, np.int64)
        self.init_pose = np.array(cfg.init_pose_buf, f32)
        # python-float scales become fp32 when they multiply an fp32 tensor
        self.s_cov = f32(cfg.scale_surface_coverage * cfg.dt)
        self.s_short = f32(cfg.scale_short_path * cfg.dt)
        self.s_term = f32(cfg.scale_termination * cfg.dt)
        self.prob_grid = np.zeros((n, g, g, g), f32)
        self.scanned_gt_grid = np.zeros((n, g, g, g), f32)
        self.pose_hist = np.tile(self.init_pose, (n, cfg.stack, 1)).astype(f32)  # oldest -> newest
        self.rgb_hist = np.zeros((n, cfg.rgb_k, cfg.rgb_h, cfg.rgb_w), f32)
        self.prev_ratio = np.zeros(n, f32)
        self.episode_length_buf = np.zeros(n, np.int64)
        self.pending_reset = np.zeros(n, np.uint8)  # grids to zero before the next update
        self.extras_time_outs = np.zeros(n, bool)
        self.actions = np.tile(self.init_action, (n, 1))

    # -- helpers ---------------------------------------------------------------
    def _poses(self, actions):
        return (actions.astype(f32) * self.action_unit + self.action_low).astype(f32)

    def _observe(self, depth_raw, seg_raw, rgba, c2w, poses):
        cfg = self.cfg
        dp, sp = orc.post_process_depth(depth_raw, seg_raw, cfg.depth_sense_dist)
        gray = orc.rgb_to_gray64(rgba, cfg.rgb_h, cfg.rgb_w)
        # update_obs_buf: deque push
        self.pose_hist = np.concatenate([self.pose_hist[:, 1:], poses[:, None, :]], axis=1)
        self.rgb_hist = np.concatenate([self.rgb_hist[:, 1:], gray], axis=1)
        tri, cov = orc.update_occ_grid(dp, sp, c2w, self.inv_intri, poses[:, :3], self.range_gt, self.voxel_size,
                                       self.grid_gt, self.prob_grid, self.scanned_gt_grid, reset_mask=self.pending_reset)
        self.pending_reset[:] = 0
        return tri, cov

    def _reward_done(self, cov):
        n = self.n
        ratio = (cov.astype(f32) / self.num_valid).astype(f32)  # _reward_surface_coverage :537
        rew = np.zeros(n, f32)
        rew = (rew + ((ratio - self.prev_ratio).astype(f32) * self.s_cov).astype(f32)).astype(f32)
        extra = np.clip(self.episode_length_buf - 30, 0, 2)  # _reward_short_path :543-545
        rew = (rew + ((-extra).astype(f32) * self.s_short).astype(f32)).astype(f32)
        if self.cfg.only_positive_rewards:
            rew = np.where(rew < 0, f32(0), rew).astype(f32)  # torch.clip(min=0.)
        time_out = self.episode_length_buf >= self.max_episode_length
        reset = time_out | (ratio > f32(self.cfg.coverage_threshold))  # collisions: none in a replay feed
        rew = (rew + ((reset & ~time_out).astype(f32) * self.s_term).astype(f32)).astype(f32)
        return rew, reset, time_out, ratio

    def _flat_obs(self, tri):
        n = self.n
        return np.concatenate([self.pose_hist.reshape(n, -1), tri.reshape(n, -1), self.rgb_hist.reshape(n, -1)], axis=1)

    def _finish(self, tri, cov):
        rew, reset, time_out, ratio = self._reward_done(cov)
        obs = self._flat_obs(tri)  # built BEFORE reset_idx mutates the buffers
        self.prev_ratio = np.where(reset, f32(0), ratio).astype(f32)
        if reset.any():
            self.extras_time_outs = time_out.copy()
            self.pose_hist[reset] = self.init_pose
            self.rgb_hist[reset] = 0
            self.actions[reset] = self.init_action
            self.episode_length_buf[reset] = 0
            self.pending_reset[reset] = 1
        return obs, rew, reset, {"time_outs": self.extras_time_outs.copy(), "coverage": ratio}

    # -- protocol ----------------------------------------------------------------
    def reset(self, depth_raw, seg_raw, rgba, c2w):
        """Env_Train_GenNBV.reset :229-244 (frame = what the simulator renders at the init pose)."""
        n = self.n
        self.pose_hist[:] = self.init_pose
        self.rgb_hist[:] = 0
        self.prev_ratio[:] = 0
        self.actions[:] = self.init_action
        self.pending_reset[:] = 1
        self.episode_length_buf[:] = 0
        self.extras_time_outs = np.zeros(n, bool)
        self.actions = np.clip(self.actions, self.clip_low, self.clip_up)
        poses = self._poses(self.actions)
        self.episode_length_buf += 1
        tri, cov = self._observe(depth_raw, seg_raw, rgba, c2w, poses)
        obs, _, _, _ = self._finish(tri, cov)
        return obs

    def step(self, actions, depth_raw, seg_raw, rgba, c2w):
        self.actions = np.clip(np.asarray(actions, np.int64), self.clip_low, self.clip_up)
        self.actions[self.episode_length_buf == 0] = self.init_action  # :249-253
        poses = self._poses(self.actions)
        self.episode_length_buf += 1
        tri, cov = self._observe(depth_raw, seg_raw, rgba, c2w, poses)
        return self._finish(tri, cov)
