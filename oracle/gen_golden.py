"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (this container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python oracle/gen_golden.py [names...]

Each fixture stores seeded inputs and the outputs the reference's own Python
(/root/reference, imported through oracle/ref_harness.py with stubbed
third-party modules) produced for them on CPU (torch 2.10 CPU here; the
reference pins torch 1.13.1+cu117 -- noted in DESIGN.md).  Fixtures are data
only; no reference source is stored.

  F1_postproc   A1  env_train_base.py:513-534 (depth / seg branch; rgb branch unpinned)
  F2_backproj   A2  env_train_gennbv.py:494-533 (+A3 per-point indices, A4 pose idx)
  F4_bresenham  A5  utils.py:43-197 kernel text compiled by oracle/build_ref.py
  F5_envstep_*  A1-A9 whole Env_Train_GenNBV.step() on a fake simulator fed by the
                synthetic feed: obs dict, flat obs, rewards, dones, grids per step
  F8_gae        C2 / C-alt  buffers.py:706-724, rsl_rl/storage/rollout_storage.py:130-144
  F7/F9         encoder + PPO (see gen_golden_ppo.py)
"""
from __future__ import annotations

import hashlib
import os
import sys
import types
from collections import deque

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gennbv_amd.env import synthetic as S  # noqa: E402
from gennbv_amd.env.config import TaskConfig  # noqa: E402
import ref_harness  # noqa: E402
import oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DEPTH_Q = 512.0  # depth quantum: raw depth = -(q / 512) exactly representable in fp32


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def quantize_depth(depth_raw: torch.Tensor):
    """-> (uint16 codes, exact fp32 depth).  65535 encodes -inf (ray hits nothing)."""
    t = -depth_raw
    q = torch.where(torch.isinf(t), torch.full_like(t, 65535.0), torch.clamp(torch.round(t * DEPTH_Q), 0, 65534))
    q = q.to(torch.int32)
    return q.numpy().astype(np.uint16), dequantize_depth(q.numpy().astype(np.uint16))


def dequantize_depth(q: np.ndarray) -> torch.Tensor:
    t = torch.from_numpy(q.astype(np.float32)) / DEPTH_Q
    t = -t
    t[torch.from_numpy(q == 65535)] = -float("inf")
    return t


def pack_bits(a: np.ndarray) -> np.ndarray:
    return np.packbits(a.astype(bool).reshape(a.shape[0], -1), axis=1)


# --------------------------------------------------------------------------- #
# a reference Env_Train_GenNBV instance on a fake simulator
# --------------------------------------------------------------------------- #
class _Obj:
    pass


def make_ref_env(ref, cfg: TaskConfig, scene: S.Scene, max_episode_length: int):
    """object.__new__(Env_Train_GenNBV) with exactly the attributes step() touches
    (mirrors _init_buffers env_train_gennbv.py:98-202 with values from `cfg`)."""
    Env = ref.env_train.Env_Train_GenNBV
    env = object.__new__(Env)
    n, g = scene.grid_gt.shape[0], cfg.grid_size
    dev = "cpu"
    env.device = dev
    env.num_envs = n
    env.grid_size = g
    env.gym = ref_harness._Anything("gym")
    env.sim = ref_harness._Anything("sim")
    env.viewer = None
    env.enable_viewer_sync = False
    env.debug_viz = False
    c = _Obj()
    c.return_visual_observation = True
    c.visual_input = _Obj()
    c.visual_input.normalization = True
    c.visual_input.camera_height, c.visual_input.camera_width = cfg.camera_height, cfg.camera_width
    c.visual_input.horizontal_fov = cfg.horizontal_fov
    c.visual_input.stack = cfg.stack
    c.normalization = _Obj()
    c.normalization.init_action = cfg.init_action
    c.normalization.init_pose_buf = cfg.init_pose_buf
    c.rewards = _Obj()
    c.rewards.only_positive_rewards = cfg.only_positive_rewards
    c.termination = _Obj()
    c.termination.max_step_done = True
    c.terrain = _Obj()
    c.terrain.curriculum = False
    c.commands = _Obj()
    c.commands.curriculum = False
    c.env = _Obj()
    c.env.send_timeouts = True
    env.cfg = c
    env.max_episode_length = max_episode_length
    env.max_episode_length_s = cfg.episode_length_s
    env.dt = cfg.dt
    # scene / GT  (env_train_gennbv.py:56-96)
    env.grid_gt = scene.grid_gt.clone()
    env.range_gt = scene.range_gt.clone()
    env.voxel_size_gt = scene.voxel_size.clone()
    env.num_valid_voxel_gt = scene.num_valid_voxel_gt.clone()
    env.env_origins = scene.env_origins.clone()
    # buffers (env_train_gennbv.py:121-200)
    env.buffer_size = cfg.stack
    env.actions = torch.tensor(cfg.init_action, dtype=torch.long).repeat(n, 1)
    env.action_unit = torch.tensor(cfg.action_unit)
    env.action_size = 6
    env.action_low_world = torch.tensor(cfg.clip_pose_low)
    env.clip_pose_idx_low = torch.tensor(cfg.clip_pose_idx_low, dtype=torch.int64)
    env.clip_pose_idx_up = torch.tensor(cfg.clip_pose_idx_up, dtype=torch.int64)
    pose_buf = torch.tensor(cfg.init_pose_buf, dtype=torch.float).repeat(n, 1)
    env.pose_buf = deque(maxlen=env.buffer_size)
    env.pose_buf.extend(env.buffer_size * [pose_buf])
    env.reward_ratio_buf = deque(maxlen=max(env.buffer_size, 2))
    env.reward_ratio_buf.extend(max(env.buffer_size, 2) * [torch.zeros(n)])
    env.collision_buf = torch.ones(n, dtype=torch.long)
    env.blender2opencv = torch.FloatTensor([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
    env.inv_intri = torch.linalg.inv(Env.get_camera_intrinsics(env)).to(torch.float32)
    H, W = cfg.camera_height, cfg.camera_width
    xs = torch.linspace(0, W - 1, W, dtype=torch.float32)
    ys = torch.linspace(0, H - 1, H, dtype=torch.float32)
    ys, xs = torch.meshgrid(ys, xs, indexing="ij")
    ncp = torch.stack([xs, ys], dim=-1)
    env.norm_coord_pixel = torch.concat((ncp, torch.ones_like(ncp[..., :1])), dim=-1).view(-1, 3)
    env.scanned_gt_grid = torch.zeros(n, g, g, g)
    env.prob_grid = torch.zeros(n, g, g, g)
    env.occ_grids_tri_cls = torch.zeros(n, g, g, g)
    env.k, env.rgb_h, env.rgb_w = cfg.rgb_k, cfg.rgb_h, cfg.rgb_w
    env.rgb_buf = deque(maxlen=env.k)
    env.rgb_buf.extend(env.k * [torch.zeros((n, 1, env.rgb_h, env.rgb_w))])
    # base-task state (legged_gym/env/base/base_task.py, drone_robot.py:660-691)
    env.episode_length_buf = torch.zeros(n, dtype=torch.long)
    env.reset_buf = torch.ones(n, dtype=torch.long)
    env.time_out_buf = torch.zeros(n, dtype=torch.bool)
    env.rew_buf = torch.zeros(n)
    env.extras = {}
    env.reward_scales = {"surface_coverage": cfg.scale_surface_coverage * cfg.dt,
                         "short_path": cfg.scale_short_path * cfg.dt,
                         "termination": cfg.scale_termination * cfg.dt}
    env.reward_names = ["surface_coverage", "short_path"]
    env.reward_functions = [env._reward_surface_coverage, env._reward_short_path]
    env.episode_sums = {k: torch.zeros(n) for k in env.reward_scales}
    env.contact_forces = torch.zeros(n, 6, 3)
    env.termination_contact_indices = torch.tensor([0, 2, 3, 4, 5])
    env.rewbuffer, env.lenbuffer = deque(maxlen=100), deque(maxlen=100)
    env.cur_reward_sum = torch.zeros(n)
    env.cur_episode_length = torch.zeros(n)
    env.camera_handles = list(range(n))
    # fake simulator hooks
    env._feed = {}
    env.render = lambda *a, **k: None
    env.set_state = lambda *a, **k: None
    env._reset_root_states = lambda *a, **k: None
    env.get_camera_view_matrix = lambda: env._feed["view"].numpy()
    # torchvision is absent: the rgb branch is UNPINNED -- plug the oracle's restatement
    def _gray(rgb_images):  # [N,3,64,64] uint8 (after nearest interpolate)
        r = rgb_images.float()
        v = 0.2989 * r[:, 0:1] + 0.587 * r[:, 1:2] + 0.114 * r[:, 2:3]
        return v.to(torch.uint8)
    ref.env_base.rgb_to_grayscale = _gray
    return env


def feed_frame(env, depth_raw, seg_raw, rgba, view):
    n = env.num_envs
    env.depth_cam_tensors = [depth_raw[i] for i in range(n)]
    env.seg_cam_tensors = [seg_raw[i] for i in range(n)]
    env.rgb_cam_tensors = [rgba[i] for i in range(n)]
    env._feed["view"] = view


def patch_bresenham(ref):
    """bresenham3D_pycuda needs pycuda; route it to the compiled REFERENCE kernel text."""
    def bres(pts_source, pts_target, map_size):
        if isinstance(map_size, list):
            map_size = map_size[0]
        src = pts_source.int().contiguous().numpy().reshape(-1)[:3]
        tgt = pts_target.int().contiguous().numpy()
        traj, lens = orc.ref_bresenham3d(src, tgt, int(map_size))
        mask = np.arange(3 * map_size)[None, :] < lens[:, None]
        return torch.from_numpy(traj[mask].reshape(-1, 3)).to(torch.long)
    ref.env_train.bresenham3D_pycuda = bres


# --------------------------------------------------------------------------- #
def gen_envstep(ref, name, n, h, w, g, num_frames, num_steps, max_ep_len, seed, keep_steps, special=True):
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=seed)
    frames = S.make_frames(scene, cfg, num_frames, seed=seed)
    depth_q, depth_f, segs, rgbas, views, acts = [], [], [], [], [], []
    for f in frames:
        q, d = quantize_depth(f.depth_raw)
        depth_q.append(q); depth_f.append(d)
        segs.append(f.seg_raw.numpy().astype(np.uint8)); rgbas.append(f.rgba.numpy())
        views.append(f.view.numpy()); acts.append(f.actions.numpy())
    # special raw-depth values in frame 0 to pin A1's nan_to_num / clamp / abs
    sp_idx = np.zeros((0,), np.int64); sp_val = np.zeros((0,), np.float32)
    if special:
        rs = np.random.RandomState(seed)
        sp_idx = rs.choice(n * h * w, size=24, replace=False).astype(np.int64)
        sp_val = np.array([np.nan, np.inf, -np.inf, -75.5, -50.0, -49.99, 0.0, -0.0] * 3, np.float32)
        flat = depth_f[0].view(-1)
        flat[torch.from_numpy(sp_idx)] = torch.from_numpy(sp_val)
    env = make_ref_env(ref, cfg, scene, max_ep_len)
    patch_bresenham(ref)
    Env = ref.env_train.Env_Train_GenNBV
    init_len = (torch.arange(n) * 3) % max(max_ep_len - 1, 1)
    out = dict(n=n, h=h, w=w, g=g, num_frames=num_frames, num_steps=num_steps, max_episode_length=max_ep_len,
               seed=seed, depth_q=np.stack(depth_q), seg=np.stack(segs), rgba=np.stack(rgbas), view=np.stack(views),
               actions=np.stack(acts), special_idx=sp_idx, special_val=sp_val,
               env_origins=scene.env_origins.numpy(), range_gt=scene.range_gt.numpy(),
               voxel_size=scene.voxel_size.numpy(), grid_gt_bits=pack_bits(scene.grid_gt.numpy()),
               num_valid_voxel_gt=scene.num_valid_voxel_gt.numpy(), init_episode_length=init_len.numpy(),
               inv_intri=env.inv_intri.numpy(), keep_steps=np.array(keep_steps))
    # reset(): reference Env.reset (:229-244) -> post_physics_step(if_reset=True)
    feed_frame(env, depth_f[0], torch.from_numpy(out["seg"][0]).float(), torch.from_numpy(out["rgba"][0]), torch.from_numpy(out["view"][0]))
    obs0 = Env.reset(env)
    wrap = ref.wrapper.flatten_observations
    out["reset_flat_obs_sha"] = sha(wrap(obs0, ["state", "grid", "state_rgb"]).numpy())
    out["reset_tri"] = obs0["grid"].numpy().astype(np.int8)
    out["reset_prob"] = env.prob_grid.numpy().copy()
    # _setup_learn overwrites episode_length_buf (base_class_grid_obs.py:471-475); we use a fixed stagger
    env.episode_length_buf = init_len.clone()
    rewards, dones, timeouts, cover, tri_sha, prob_sha, scan_sha, obs_sha, c2w_all = [], [], [], [], [], [], [], [], []
    # extras["episode"] of every step (reset_idx :424-427 rew_<name>, update_extra_episode_info base:638-639), the identity
    # of the dict object (a new one only on steps with resets: buffer entries in between alias it) and what the reference's
    # BestCKPTCallback.calculate_value computes over the ep_info_buffer the on-policy loop fills
    ep_info, ep_gen, ep_ids = [], [], {}
    from collections import deque
    ep_buffer = deque(maxlen=100)
    ep_keys = None
    keep = {}
    for s in range(num_steps):
        fi = (s + 1) % num_frames
        feed_frame(env, depth_f[fi], torch.from_numpy(out["seg"][fi]).float(), torch.from_numpy(out["rgba"][fi]),
                   torch.from_numpy(out["view"][fi]))
        a = torch.from_numpy(out["actions"][fi]).clone()
        # state BEFORE reset_idx is what the obs shows; capture grids via a hook on reset_idx
        snap = {}
        orig_reset = Env.reset_idx

        def hooked(self, env_ids, _snap=snap):
            _snap["prob"] = self.prob_grid.numpy().copy()
            _snap["scan"] = self.scanned_gt_grid.numpy().copy()
            return orig_reset(self, env_ids)
        env.reset_idx = types.MethodType(hooked, env)
        obs, rew, done, info = Env.step(env, a)
        flat = wrap(obs, ["state", "grid", "state_rgb"]).numpy()
        rewards.append(rew.numpy().copy()); dones.append(done.numpy().copy())
        timeouts.append(info["time_outs"].numpy().copy())
        cover.append(env.reward_ratio_buf[-1].numpy().copy())
        d = info["episode"]
        if ep_keys is None:
            ep_keys = sorted(d.keys())
        assert sorted(d.keys()) == ep_keys
        ep_info.append([float(d[k]) for k in ep_keys])
        ep_gen.append(ep_ids.setdefault(id(d), len(ep_ids)))
        ep_buffer.extend([d])  # base_class_grid_obs.py:491-493
        tri = obs["grid"].numpy()
        tri_sha.append(sha(tri.astype(np.float32))); prob_sha.append(sha(snap["prob"])); scan_sha.append(sha(snap["scan"]))
        obs_sha.append(sha(flat))
        if s in keep_steps:
            keep[f"tri_{s}"] = tri.astype(np.int8)
            keep[f"prob_{s}"] = snap["prob"]
            keep[f"scan_bits_{s}"] = pack_bits(snap["scan"])
            keep[f"pose_state_{s}"] = obs["state"].numpy().copy()
            keep[f"rgb_state_{s}"] = obs["state_rgb"].numpy().astype(np.uint8)
            keep[f"post_prob_{s}"] = env.prob_grid.numpy().copy()  # after reset_idx
    out.update(keep)
    out.update(rewards=np.stack(rewards), dones=np.stack(dones), time_outs=np.stack(timeouts), coverage=np.stack(cover),
               tri_sha=np.array(tri_sha), prob_sha=np.array(prob_sha), scan_sha=np.array(scan_sha),
               flat_obs_sha=np.array(obs_sha), episode_info=np.array(ep_info, np.float64), episode_keys=np.array(ep_keys),
               episode_dict_generation=np.array(ep_gen, np.int64))
    # the reference's own callback arithmetic over the (aliased) buffer entries (gennbv/callback.py:58-70)
    cb = object.__new__(ref.callback.BestCKPTCallback)
    cb.locals = {"self": types.SimpleNamespace(ep_info_buffer=ep_buffer, device="cpu")}
    out["best_ckpt_value_episode_reward"] = np.float64(ref.callback.BestCKPTCallback.calculate_value(cb, "episode_reward"))
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    print(name, "saved; mean fg", float((out["seg"] > 50).mean()), "resets", int(np.stack(dones).sum()))


def gen_backproj(ref):
    """F1 + F2 + F3/F4(pose idx): per-point pins, small image so everything is stored in full."""
    n, h, w, g = 3, 48, 64, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=5)
    fr = S.make_frames(scene, cfg, 1, seed=5)[0]
    q, depth_raw = quantize_depth(fr.depth_raw)
    rs = np.random.RandomState(5)
    sp_idx = rs.choice(n * h * w, size=16, replace=False).astype(np.int64)
    sp_val = np.array([np.nan, np.inf, -np.inf, -75.5, -50.0, -49.99, 0.0, -0.0] * 2, np.float32)
    depth_raw.view(-1)[torch.from_numpy(sp_idx)] = torch.from_numpy(sp_val)
    seg = fr.seg_raw.numpy().astype(np.uint8)
    env = make_ref_env(ref, cfg, scene, 100)
    Env = ref.env_train.Env_Train_GenNBV
    feed_frame(env, depth_raw, torch.from_numpy(seg).float(), fr.rgba, fr.view)
    Env.post_process_camera_tensor(env)
    env.poses = fr.poses.clone()
    # world points for ALL pixels: rerun the reference einsum chain without the fg compaction
    pts = Env.back_projection_fg(env)  # list of [n_i,3]
    idx_list = ref.utils.scanned_pts_to_idx_3D(pts, env.range_gt, env.voxel_size_gt, map_size=g)
    pose_idx = ref.utils.pose_coord_to_idx_3D(env.poses[:, :3].clone(), env.range_gt, env.voxel_size_gt, map_size=g)
    # out-of-grid poses for A4
    far = torch.tensor([[12.0, -9.5, 10.1], [-8.6, 8.6, -0.5], [0.0, 0.0, 10.9]])
    far_idx = ref.utils.pose_coord_to_idx_3D(far.clone(), env.range_gt, env.voxel_size_gt, map_size=g)
    extr = torch.from_numpy(env.get_camera_view_matrix())
    c2w = torch.linalg.inv(extr.transpose(-2, -1)) @ env.blender2opencv.unsqueeze(0)
    c2w[:, :3, 3] -= env.env_origins
    out = dict(n=n, h=h, w=w, g=g, depth_q=q, seg=seg, special_idx=sp_idx, special_val=sp_val,
               view=fr.view.numpy(), env_origins=scene.env_origins.numpy(), range_gt=scene.range_gt.numpy(),
               voxel_size=scene.voxel_size.numpy(), inv_intri=env.inv_intri.numpy(), c2w=c2w.numpy(),
               poses=fr.poses.numpy(), depth_processed=env.depth_processed.numpy(),
               seg_processed=env.seg_processed.numpy(), pose_idx=pose_idx.numpy(), far_poses=far.numpy(),
               far_idx=far_idx.numpy())
    for i in range(n):
        out[f"world_{i}"] = pts[i].numpy()
        out[f"uidx_{i}"] = (idx_list[i].numpy() if not isinstance(idx_list[i], list) else np.zeros((0, 3), np.int64))
    np.savez_compressed(os.path.join(GOLDEN, "F2_backproj.npz"), **out)
    print("F2_backproj saved", [len(p) for p in pts])


def gen_bresenham(ref):
    rs = np.random.RandomState(11)
    cases = {}
    for g in (16, 20, 64):
        srcs = [np.array([g // 2, g // 2, g + 5]), np.array([-7, 3, g // 3]), np.array([g + 9, g + 4, g + 11]),
                np.array([3, 4, 5]), np.array([-3 * g, -2 * g, 5 * g])]
        for si, src in enumerate(srcs):
            t = rs.randint(0, g, size=(40, 3)).astype(np.int32)
            t[0] = np.clip(src, 0, g - 1)  # (possibly) degenerate src == tgt
            t[1] = [0, 0, 0]; t[2] = [g - 1, g - 1, g - 1]
            t[3] = [np.clip(src[0], 0, g - 1), np.clip(src[1], 0, g - 1), 0]  # axis aligned
            t[4] = np.clip(src + np.array([5, 5, -5]), 0, g - 1)  # ties in the dominant axis
            t[5] = np.clip(src + np.array([-4, 4, 4]), 0, g - 1)
            traj, lens = orc.ref_bresenham3d(src.astype(np.int32), t, g)
            cases[f"g{g}_s{si}_src"] = src.astype(np.int32)
            cases[f"g{g}_s{si}_tgt"] = t
            cases[f"g{g}_s{si}_traj"] = traj.astype(np.int16)
            cases[f"g{g}_s{si}_len"] = lens
    np.savez_compressed(os.path.join(GOLDEN, "F4_bresenham.npz"), **cases)
    print("F4_bresenham saved", len(cases) // 4, "cases")


def gen_gae(ref):
    torch.manual_seed(3)
    T, N = 128, 8
    rewards = torch.randn(T, N) * 0.5
    values = torch.randn(T, N)
    starts = (torch.rand(T, N) < 0.05)
    starts[0] = True
    last_values = torch.randn(N, 1)
    dones = (torch.rand(N) < 0.3).long()
    Buf = ref.buffers.TensorRolloutBuffer_Grid_Obs
    buf = object.__new__(Buf)
    buf.buffer_size, buf.gamma, buf.gae_lambda = T, 0.99, 0.95
    buf.rewards = rewards.view(T, N, 1).clone()
    buf.values = values.view(T, N, 1).clone()
    buf.episode_starts = starts.view(T, N, 1).byte()
    buf.advantages = torch.zeros(T, N, 1)
    Buf.compute_returns_and_advantage(buf, last_values=last_values, dones=dones)
    # rsl_rl: dones stored with the transition
    St = ref.rsl_storage.RolloutStorage
    st = object.__new__(St)
    st.num_transitions_per_env = T
    st.rewards = rewards.view(T, N, 1).clone()
    st.values = values.view(T, N, 1).clone()
    rsl_dones = (torch.rand(T, N, 1) < 0.05).byte()
    st.dones = rsl_dones
    st.returns = torch.zeros(T, N, 1)
    # capture un-normalised advantages: returns - values
    St.compute_returns(st, last_values, 0.99, 0.95)
    out = dict(rewards=rewards.numpy(), values=values.numpy(), episode_starts=starts.numpy().astype(np.uint8),
               last_values=last_values.numpy().reshape(-1), dones=dones.numpy().astype(np.uint8),
               sb3_advantages=buf.advantages.numpy().reshape(T, N), sb3_returns=buf.returns.numpy().reshape(T, N),
               rsl_dones=rsl_dones.numpy().reshape(T, N), rsl_returns=st.returns.numpy().reshape(T, N),
               rsl_advantages_normalized=st.advantages.numpy().reshape(T, N))
    np.savez_compressed(os.path.join(GOLDEN, "F8_gae.npz"), **out)
    print("F8_gae saved")


def main(names):
    os.makedirs(GOLDEN, exist_ok=True)
    from build_ref import build as build_ref
    build_ref()
    ref = ref_harness.import_reference()
    torch.set_num_threads(8)
    todo = {
        "F2_backproj": lambda: gen_backproj(ref),
        "F4_bresenham": lambda: gen_bresenham(ref),
        "F8_gae": lambda: gen_gae(ref),
        # BASELINE config 0 shape: 4 envs, 240x320, 16^3 ; 30 steps, resets every <=10 steps
        "F5_envstep_c0": lambda: gen_envstep(ref, "F5_envstep_c0", 4, 240, 320, 16, 3, 30, 10, 1, [0, 1, 2, 9, 10, 29]),
        # reference default grid (20^3) on a small 400-ratio image, long enough to pin >20 decrements
        "F5_envstep_g20": lambda: gen_envstep(ref, "F5_envstep_g20", 2, 100, 100, 20, 2, 40, 100, 2, [0, 1, 20, 39]),
        # 64^3 grid (BASELINE config 1 grid) on a reduced image
        "F5_envstep_g64": lambda: gen_envstep(ref, "F5_envstep_g64", 2, 120, 160, 64, 2, 4, 100, 3, [0, 3]),
    }
    for k in (names or todo.keys()):
        todo[k]()


if __name__ == "__main__":
    main(sys.argv[1:])
