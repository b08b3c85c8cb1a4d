"""ReplayFeedEnv: Env_Train_GenNBV with Isaac Gym cut at the observation boundary.

The reference env (gennbv/env/env_train_gennbv.py + env_train_base.py) renders
depth / segmentation / RGB with Isaac Gym and then runs the state encoding.  Here a
recorded or synthetic feed supplies exactly the tensors the simulator would
(`depth_raw`, `seg_raw`, `rgba`, camera view matrix), everything after that
boundary runs on gfx950 kernels through the C-ABI, and the env speaks the
protocol the reference algorithm expects from `EnvWrapperGenNBVTrain`
(gennbv/wrapper/env_wrapper_gennbv_train.py:90-133):

    num_envs, device, observation_space (flat Box (D_obs,)), action_space
    (MultiDiscrete [81,81,51,1,13,13]), episode_length_buf (int64 [N], writable),
    max_episode_length, seed(), close(),
    reset() -> obs f32 [N, D_obs]
    step(actions int64 [N,6]) -> (obs, rewards f32 [N], dones bool [N], infos)
        infos["time_outs"] bool [N], infos["episode"] dict

The flat observation is written in place by the kernels in the wrapper's key
order [state | grid | state_rgb]; `step(..., obs_out=rows)` lets the rollout
buffer hand its own storage to the env (no assemble + copy pass).
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .. import _lib
from ..spaces import Box, MultiDiscrete
from .config import TaskConfig
from .state_encoding import OccupancyGridUpdater
from . import synthetic as S


@dataclass
class ReplayFeed:
    """A pool of recorded frames, resident in HBM: frame f of env e."""
    depth_raw: torch.Tensor  # [F,N,H,W] f32, Isaac convention (negative metres, -inf = nothing)
    seg_raw: torch.Tensor  # [F,N,H,W] f32
    rgba: Optional[torch.Tensor]  # [F,N,H,W,4] u8 or None (gray frames stay zero)
    c2w: torch.Tensor  # [F,N,4,4] f32: inv(view^T) @ blender2opencv, translation - env_origins
    cursor: int = 0

    @property
    def num_frames(self):
        return self.depth_raw.shape[0]

    def next(self):
        f = self.cursor % self.num_frames
        self.cursor += 1
        return (self.depth_raw[f], self.seg_raw[f], None if self.rgba is None else self.rgba[f], self.c2w[f])

    @staticmethod
    def from_views(depth_raw, seg_raw, rgba, view, env_origins):
        """Host plumbing of back_projection_fg (env_train_gennbv.py:512-514) done once for the
        whole recording: c2w = inv(view^T) @ blender2opencv, translation minus env_origins."""
        f, n = view.shape[0], view.shape[1]
        c2w = S.c2w_from_view(view.reshape(f * n, 4, 4), env_origins.repeat(f, 1)).reshape(f, n, 4, 4)
        return ReplayFeed(depth_raw.contiguous(), seg_raw.contiguous(), None if rgba is None else rgba.contiguous(),
                          c2w.contiguous())

    @staticmethod
    def from_file(path: str, device, first: int = 0, count: Optional[int] = None):
        """Frames of a recorded-feed container (env/feed_file.py), decoded on `device`."""
        from . import feed_file
        return feed_file.load_feed(feed_file.FeedFile(path), device, first, count)

    @staticmethod
    def synthetic(scene: S.Scene, cfg: TaskConfig, num_frames: int, seed: int = 1, with_rgba: bool = True):
        frames = S.make_frames(scene, cfg, num_frames, seed=seed, with_rgba=with_rgba)
        return ReplayFeed.from_views(torch.stack([f.depth_raw for f in frames]), torch.stack([f.seg_raw for f in frames]),
                                     torch.stack([f.rgba for f in frames]) if with_rgba else None,
                                     torch.stack([f.view for f in frames]), scene.env_origins)


class ReplayFeedEnv:
    @classmethod
    def from_file(cls, cfg: TaskConfig, path: str, device="cuda:0", max_episode_length: Optional[int] = None):
        """Env over a recorded-feed container: scene block + frames (env/feed_file.py)."""
        import dataclasses

        from . import feed_file
        ff = feed_file.FeedFile(path)
        cfg = dataclasses.replace(cfg, camera_height=ff.height, camera_width=ff.width, grid_size=ff.grid_size)
        # the recording's own inverse intrinsics (a different FOV than cfg's must not be back-projected with cfg's K)
        kinv = torch.from_numpy(np.array(ff.scene("inv_intrinsics"), dtype=np.float32))
        return cls(cfg, feed_file.load_scene(ff, "cpu"), feed_file.load_feed(ff, device), device, max_episode_length,
                   inv_intrinsics=kinv)

    def __init__(self, cfg: TaskConfig, scene: S.Scene, feed: ReplayFeed, device="cuda:0",
                 max_episode_length: Optional[int] = None, inv_intrinsics: Optional[torch.Tensor] = None):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.GennbvHipError("ReplayFeedEnv runs on the GPU only (no CPU fallback)")
        self.feed = feed
        n = scene.grid_gt.shape[0]
        self.num_envs = n
        self.grid_size = cfg.grid_size
        self.max_episode_length = int(cfg.max_episode_length if max_episode_length is None else max_episode_length)
        self.max_episode_length_s = cfg.episode_length_s
        dev = self.device
        self.updater = OccupancyGridUpdater(n, cfg.grid_size, cfg.camera_height, cfg.camera_width,
                                            S.inverse_intrinsics(cfg.camera_height, cfg.camera_width, cfg.horizontal_fov)
                                            if inv_intrinsics is None else inv_intrinsics,
                                            scene.range_gt, scene.voxel_size, scene.grid_gt, dev, cfg.depth_sense_dist,
                                            max_steps_between_resets=self.max_episode_length + 1)  # (+1: the reset observation)
        self.num_valid_voxel_gt = scene.num_valid_voxel_gt.to(dev, torch.float32).contiguous()
        # spaces (update_observation_space :459-492 flattened by the wrapper :59-88)
        self.action_space = MultiDiscrete(cfg.action_nvec)
        self.observation_space = Box(-np.inf, np.inf, shape=(cfg.obs_dim,), dtype=np.float32)
        # lattice constants for the kernels
        lat = _lib.GnbvLattice()
        for i in range(6):
            lat.clip_low[i], lat.clip_up[i] = cfg.clip_pose_idx_low[i], cfg.clip_pose_idx_up[i]
            lat.init_action[i] = cfg.init_action[i]
            lat.action_unit[i] = float(np.float32(cfg.action_unit[i]))
            lat.pose_low[i] = float(np.float32(cfg.clip_pose_low[i]))
            lat.init_pose[i] = float(np.float32(cfg.init_pose_buf[i]))
        self._lat = lat
        # state tensors
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        self.episode_length_buf = z(n, dt=torch.int64)
        self.actions = torch.tensor(cfg.init_action, dtype=torch.int64, device=dev).repeat(n, 1)
        self.poses = z(n, 6)
        self.pose_hist = torch.tensor(cfg.init_pose_buf, dtype=torch.float32, device=dev).repeat(n, cfg.stack, 1)
        self.gray_prev = z(n, cfg.rgb_h * cfg.rgb_w)
        self.prev_ratio = z(n)
        self.rew_buf = z(n)
        self.reset_buf = z(n, dt=torch.uint8)
        # `flag_views` (off by default; collect_rollouts of PPO_Grid_Obs turns it on): step() hands out `dones` and infos["time_outs"] as
        # bool VIEWS of the kernels' 0 / 1 bytes instead of `.bool()` copies (two element-wise launches per env step).  `dones` of step t is
        # still needed after step t + 1 has run (it is step t + 1's episode_starts): two buffers take turns.  time_outs is read inside the
        # step that produced it (the time-out bootstrap) and is STATEFUL on the device (the reference only refreshes it on steps with a
        # reset): one buffer, and a reader that keeps infos["time_outs"] across env steps must copy it.
        self.flag_views = False
        self.fused_observe = True  # pre-step + both observation slices as one launch (False: the three separate entry points, kept covered by tests/test_envstep_gpu.py)
        self._reset_bufs = (self.reset_buf, z(n, dt=torch.uint8))
        self._reset_turn = 0
        self.reset_mask = torch.ones(n, dtype=torch.uint8, device=dev)
        self.time_out_buf = z(n, dt=torch.uint8)
        self.extras_time_outs = z(n, dt=torch.uint8)
        self.coverage_ratio = z(n)
        self.episode_sums = z(3, n)
        self.cur_reward_sum = z(n)
        self.cur_episode_length = z(n)
        self.ring_len = 100
        self.ring_reward = z(self.ring_len)
        self.ring_length = z(self.ring_len)
        self.ring_state = z(1, dt=torch.int64)
        self._zero_rgba = None
        self._obs = torch.zeros(n, cfg.obs_dim, dtype=torch.float32, device=dev)
        p = _lib.GnbvEnvPost()
        p.n, p.only_positive, p.max_episode_length = n, int(cfg.only_positive_rewards), self.max_episode_length
        # a Python-float scale multiplying an fp32 tensor is rounded to fp32 first
        p.scale_cov = float(np.float32(cfg.scale_surface_coverage * cfg.dt))
        p.scale_short = float(np.float32(cfg.scale_short_path * cfg.dt))
        p.scale_term = float(np.float32(cfg.scale_termination * cfg.dt))
        p.coverage_threshold = float(np.float32(cfg.coverage_threshold))
        p.coverage_count = self.updater.coverage_count.data_ptr()
        p.num_valid_voxel_gt = self.num_valid_voxel_gt.data_ptr()
        p.prev_ratio = self.prev_ratio.data_ptr()
        p.rewards, p.dones = self.rew_buf.data_ptr(), self.reset_buf.data_ptr()
        p.reset_mask, p.step_time_out = self.reset_mask.data_ptr(), self.time_out_buf.data_ptr()
        p.extras_time_outs, p.coverage_ratio = self.extras_time_outs.data_ptr(), self.coverage_ratio.data_ptr()
        p.episode_sums, p.cur_reward_sum = self.episode_sums.data_ptr(), self.cur_reward_sum.data_ptr()
        p.cur_episode_length = self.cur_episode_length.data_ptr()
        p.ring_reward, p.ring_length = self.ring_reward.data_ptr(), self.ring_length.data_ptr()
        p.ring_state, p.ring_len = self.ring_state.data_ptr(), self.ring_len
        # extras["episode"] snapshots: k_env_post_step writes this step's (mean reward, mean length) into slot
        # step % H; the dict handed out with the step reads its own slot on demand (no per-step host sync)
        self._ep_hist = 1024
        self.episode_info_hist = z(self._ep_hist, 6, dt=torch.float64)  # per step: generation, 2 deque means, 3 rew_<name>
        self.episode_state = z(4, dt=torch.float64)
        p.episode_state, p.max_episode_length_s = self.episode_state.data_ptr(), float(np.float32(self.max_episode_length_s))
        self._ep_step = 0
        self._ep_cache = (-1, None)
        self._ep_live = collections.deque()
        self._post = p
        self.extras = {}

    # reference attribute names for the grids
    @property
    def prob_grid(self):
        return self.updater.prob_grid

    @property
    def scanned_gt_grid(self):
        return self.updater.scanned_gt_grid

    @property
    def grid_gt(self):
        return self.updater.grid_gt

    def seed(self, seed=None):
        return [seed]

    def close(self):
        pass

    # ------------------------------------------------------------------------
    def _observe_and_finish(self, actions_in: torch.Tensor, obs_out: Optional[torch.Tensor], grid_i8_out: Optional[torch.Tensor] = None):
        cfg, n, lib = self.cfg, self.num_envs, self.lib
        st = _lib.stream_ptr(self.device)
        obs = self._obs if obs_out is None else obs_out
        # compact rows [state | state_rgb]: the grid goes to `grid_i8_out` only (coded update)
        compact = obs.shape == (n, cfg.obs_dim - cfg.grid_dim)
        if compact and (grid_i8_out is None or not self.updater.coded):
            raise _lib.GennbvHipError("compact observation rows need grid_i8_out and the coded grid update")
        assert (compact or obs.shape == (n, cfg.obs_dim)) and obs.dtype == torch.float32 and obs.stride(1) == 1
        stride = obs.stride(0)
        depth_raw, seg_raw, rgba, c2w = self.feed.next()
        # step(): clip, forced init action, poses; episode_length_buf += 1
        self._post.episode_length_buf = self.episode_length_buf.data_ptr()  # the algorithm may have replaced the tensor
        rgb_off = cfg.state_dim + (0 if compact else cfg.grid_dim)
        if rgba is None:
            if self._zero_rgba is None:
                self._zero_rgba = torch.zeros(n, cfg.camera_height, cfg.camera_width, 4, dtype=torch.uint8, device=self.device)
            rgba = self._zero_rgba
        fused_observe = self.fused_observe and getattr(lib, "gnbv_env_observe", None) is not None
        if fused_observe:
            # round 5: the three launches below as one (same arithmetic, bit-identical: tests/test_envstep_gpu.py)
            _lib.check(lib.gnbv_env_observe(actions_in.data_ptr(), C.byref(self._lat), self.episode_length_buf.data_ptr(), n, self.actions.data_ptr(),
                                            self.poses.data_ptr(), self.pose_hist.data_ptr(), self.reset_mask.data_ptr(), cfg.stack, obs.data_ptr(), stride,
                                            rgba.data_ptr(), self.gray_prev.data_ptr(), cfg.camera_height, cfg.camera_width, cfg.rgb_h, cfg.rgb_w,
                                            obs.data_ptr() + 4 * rgb_off, st), "gnbv_env_observe")
        else:
            self._observe_three_launches(actions_in, obs, stride, rgba, rgb_off, st)
        # obs["grid"]: tri-class grid straight into the observation rows
        if compact:
            self.updater.update(depth_raw, seg_raw, c2w, self.poses, reset_mask=self.reset_mask, tri_i8_out=grid_i8_out, fp32_out=False)
        else:
            self.updater.update(depth_raw, seg_raw, c2w, self.poses, reset_mask=self.reset_mask,
                                tri_out=obs[:, cfg.state_dim:], tri_row_stride=stride,
                                tri_i8_out=grid_i8_out if self.updater.coded else None)
        # rewards / termination / reset bookkeeping
        self._ep_step += 1
        self._post.episode_info = self.episode_info_hist[self._ep_step % self._ep_hist].data_ptr()
        _lib.check(lib.gnbv_env_post_step(C.byref(self._post), st), "gnbv_env_post_step")
        return obs

    def _observe_three_launches(self, actions_in, obs, stride, rgba, rgb_off, st):
        """step()'s head and the two small observation slices as the three separate launches (`fused_observe = False`)."""
        cfg, n, lib = self.cfg, self.num_envs, self.lib
        _lib.check(lib.gnbv_env_pre_step(actions_in.data_ptr(), C.byref(self._lat), self.episode_length_buf.data_ptr(), n,
                                         self.actions.data_ptr(), self.poses.data_ptr(), st), "gnbv_env_pre_step")
        # (Round 5 ran the two small observation kernels below on a second stream beside the voxel update, joined in front of the
        # post-step kernel: the env step +1.9 ... +5.4 us, the voxel update +4.9 us -- they take slots from k_hit_list's single round of
        # workgroups.  profiles/r05_ab_rollout_obs_overlap.json)
        # obs["state"]
        _lib.check(lib.gnbv_env_obs_state(self.pose_hist.data_ptr(), self.poses.data_ptr(), self.reset_mask.data_ptr(),
                                          C.byref(self._lat), n, cfg.stack, obs.data_ptr(), stride, st), "gnbv_env_obs_state")
        # obs["state_rgb"]
        _lib.check(lib.gnbv_env_obs_rgb(rgba.data_ptr(), self.gray_prev.data_ptr(), self.reset_mask.data_ptr(), n,
                                        cfg.camera_height, cfg.camera_width, cfg.rgb_h, cfg.rgb_w,
                                        obs.data_ptr() + 4 * rgb_off, stride, st), "gnbv_env_obs_rgb")

    @property
    def supports_grid_i8(self) -> bool:
        """step(..., grid_i8_out=rows) also writes an int8 copy of the tri-class grid (coded update only)."""
        return bool(self.updater.coded)

    def reset(self, obs_out: Optional[torch.Tensor] = None, grid_i8_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Env_Train_GenNBV.reset (:229-244): reset every env, observe at the init pose."""
        n = self.num_envs
        self.episode_length_buf.zero_()
        self.reset_mask.fill_(1)
        self.prev_ratio.zero_()
        self.extras_time_outs.zero_()
        self.gray_prev.zero_()
        # reset_idx(all envs) (:231): a new extras["episode"] dict with rew_<name> = mean(episode_sums) / episode_length_s
        self.episode_state[0] += 1
        self.episode_state[1:] = (self.episode_sums.mean(dim=1) / np.float32(self.max_episode_length_s)).double()
        self.episode_sums.zero_()
        init = torch.tensor(self.cfg.init_action, dtype=torch.int64, device=self.device).repeat(n, 1)
        return self._observe_and_finish(init, obs_out, grid_i8_out)

    def step(self, actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None, grid_i8_out: Optional[torch.Tensor] = None):
        _lib.require_cuda(actions)
        a = actions.to(torch.int64).contiguous()
        views = self.flag_views
        if views:
            self._reset_turn ^= 1
            self.reset_buf = self._reset_bufs[self._reset_turn]
            self._post.dones = self.reset_buf.data_ptr()
        obs = self._observe_and_finish(a, obs_out, grid_i8_out)
        self.extras["time_outs"] = self.extras_time_outs.view(torch.bool) if views else self.extras_time_outs.bool()
        info = _LazyEpisodeInfo(self, self._ep_step)
        self.extras["episode"] = info
        # dicts that are still referenced when their snapshot slot is about to be overwritten keep their last values (the
        # reference hands out plain dicts that never fail); the common case -- nobody holds a dict for ~1000 env steps --
        # costs one weakref per step and no read-back
        live = self._ep_live
        live.append(weakref.ref(info))
        while live:
            d = live[0]()
            if d is not None and self._ep_step - d._step < self._ep_hist - 2:
                break
            live.popleft()
            if d is not None:
                d._freeze()
        return obs, self.rew_buf, (self.reset_buf.view(torch.bool) if views else self.reset_buf.bool()), self.extras

    # ------------------------------------------------------------------------
    REWARD_NAMES = ("surface_coverage", "short_path", "termination")  # cfg.rewards.scales order = episode_sums order

    def episode_info(self, step: Optional[int] = None):
        """extras["episode"] of the reference as a reader of the entry handed out at env step `step` sees it NOW
        (default: the latest step).  The reference creates a new dict only on steps where some env resets
        (reset_idx :424-428) and otherwise keeps mutating the same object (update_extra_episode_info base:629-639), so
        an older buffer entry shows the values of the LAST step its dict was alive.  Resolved on demand from the
        per-step device snapshots the post-step kernel wrote (one read-back per env step at most; the reference pays
        a .cpu() per step)."""
        step = self._ep_step if step is None else step
        if self._ep_step - step >= self._ep_hist or step < 1:
            raise _lib.GennbvHipError("episode info of a step outside the snapshot history")
        if self._ep_cache[0] != self._ep_step:
            self._ep_cache = (self._ep_step, self.episode_info_hist.cpu().numpy())
        hist = self._ep_cache[1]
        gen = hist[step % self._ep_hist][0]
        last = step
        while last < self._ep_step and hist[(last + 1) % self._ep_hist][0] == gen:
            last += 1
        row = hist[last % self._ep_hist]
        out = {"rew_" + k: float(row[3 + i]) for i, k in enumerate(self.REWARD_NAMES)}
        out["episode_reward"], out["episode_length"] = float(row[1]), float(row[2])
        return out


class _LazyEpisodeInfo(dict):
    """The reference's extras["episode"] dict handed out with ONE env step; filled from the device snapshots when
    something reads it (every dict accessor fills first; see ReplayFeedEnv.episode_info for the aliasing rule)."""

    def __init__(self, env, step):
        super().__init__()
        self._env, self._step, self._seen = env, step, -1

    def _freeze(self):
        """Resolve once more and detach from the env: from here on a plain dict with the last values."""
        self._fill()
        self._env = None

    def _fill(self):
        if self._env is None:
            return
        if self._seen != self._env._ep_step:  # (the reference keeps mutating the dict while it is alive: refresh per env step)
            self._seen = self._env._ep_step
            super().update(self._env.episode_info(self._step))

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __contains__(self, k):
        self._fill()
        return super().__contains__(k)

    def __len__(self):
        self._fill()
        return super().__len__()

    def keys(self):
        self._fill()
        return super().keys()

    def values(self):
        self._fill()
        return super().values()

    def items(self):
        self._fill()
        return super().items()

    def get(self, k, d=None):
        self._fill()
        return super().get(k, d)

    def __repr__(self):
        self._fill()
        return super().__repr__()

    def __eq__(self, o):
        self._fill()
        return super().__eq__(o)

    def copy(self):
        self._fill()
        return super().copy()

    __hash__ = None

    def __bool__(self):
        self._fill()
        return super().__len__() > 0
