"""ReplayFeedEvalEnv: Env_Eval_GenNBV (gennbv/env/env_eval_gennbv.py) over a recorded / synthetic feed.

The evaluation env is the training env plus the reconstruction-accuracy metric: every step appends the
back-projected foreground points of each env to a per-episode list (:156-164); when an env finishes, the list is
rounded to 1 cm, de-duplicated and compared with the env's GT point cloud by Chamfer distance, x100 (:253-262);
`reset_idx` then empties the list (:321-322).  `reset()` and `step()` return the reference's 5-tuple
`(obs, rewards, dones, infos, ratios_accuracy)` (:104-111, :150) that `evaluate_policy_grid_obs` consumes;
`ratios_accuracy[str(env)]` keeps the FIRST finished episode of each env like the reference (:262-263).

The back projection is the standalone A1/A2 kernels (`gnbv_post_process_depth`, `gnbv_back_projection`), the metric
`gnbv_chamfer_distance` (gennbv_amd/eval/metrics.py).  GT clouds: the reference loads one `.pt` per scene (:95-101);
without files the centres of the occupied GT voxels are used.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import utils as U
from ..eval import metrics as M
from . import synthetic as S
from .config import TaskConfig
from .replay_feed import ReplayFeed, ReplayFeedEnv


class _RecordingFeed:
    """Feed proxy that remembers the frame it handed out last."""

    def __init__(self, feed: ReplayFeed):
        self._feed, self.last = feed, None

    def next(self):
        self.last = self._feed.next()
        return self.last

    def __getattr__(self, k):
        return getattr(self._feed, k)


def gt_cloud_from_grid(grid_gt: torch.Tensor, range_gt: torch.Tensor, voxel_size: torch.Tensor) -> List[torch.Tensor]:
    """Centres of the occupied GT voxels, per env: voxel i along an axis covers [min - v/2 + i v, min + v/2 + i v)
    (scanned_pts_to_idx_3D, gennbv/utils.py:230-270), so its centre is min + i v.  range_gt = (xmax,xmin,ymax,ymin,zmax,zmin)."""
    out = []
    for e in range(grid_gt.shape[0]):
        idx = torch.nonzero(grid_gt[e] > 0).to(torch.float32)
        mins = range_gt[e, [1, 3, 5]].to(idx.device, torch.float32)
        out.append(mins + idx * voxel_size[e].to(idx.device, torch.float32))
    return out


class ReplayFeedEvalEnv(ReplayFeedEnv):
    def __init__(self, cfg: TaskConfig, scene: S.Scene, feed: ReplayFeed, device="cuda:0", max_episode_length: Optional[int] = None,
                 pc_gt: Optional[List[torch.Tensor]] = None):
        super().__init__(cfg, scene, _RecordingFeed(feed), device, max_episode_length)
        self.pc_gt = [p.to(self.device, torch.float32).contiguous() for p in
                      (pc_gt if pc_gt is not None else gt_cloud_from_grid(scene.grid_gt, scene.range_gt, scene.voxel_size))]
        assert len(self.pc_gt) == self.num_envs
        self._inv_intri = S.inverse_intrinsics(cfg.camera_height, cfg.camera_width, cfg.horizontal_fov)
        self.pts_target_list: List[List[torch.Tensor]] = [[] for _ in range(self.num_envs)]
        self.ratios_accuracy = {}

    def _accumulate_and_score(self) -> None:
        depth_raw, seg_raw, _, c2w = self.feed.last
        depth, seg = U.post_process_depth(depth_raw, seg_raw, self.cfg.depth_sense_dist)
        pts = U.back_projection_fg(depth, seg, c2w, self._inv_intri)  # list of [n_i, 3]
        for e in range(self.num_envs):
            self.pts_target_list[e].append(pts[e])
        for e in torch.nonzero(self.reset_buf).flatten().tolist():
            cloud = torch.cat(self.pts_target_list[e], 0)
            if cloud.shape[0] > 0 and str(e) not in self.ratios_accuracy:
                self.ratios_accuracy[str(e)] = float(M.reconstruction_accuracy_cm(cloud, self.pc_gt[e]))
            self.pts_target_list[e] = []  # reset_idx (:321-322)

    def reset(self, obs_out=None):
        self.pts_target_list = [[] for _ in range(self.num_envs)]
        obs = super().reset(obs_out)
        self._accumulate_and_score()
        self.extras["time_outs"] = self.extras_time_outs.bool()
        return obs, self.rew_buf, self.reset_buf.bool(), self.extras, self.ratios_accuracy

    def step(self, actions: torch.Tensor, obs_out=None):
        obs, rew, dones, infos = super().step(actions, obs_out)
        self._accumulate_and_score()
        return obs, rew, dones, infos, self.ratios_accuracy
