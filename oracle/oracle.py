"""ctypes front-end of the CPU oracle (oracle/oracle.c) + the compiled reference kernel.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the gennbv_amd product package.
All arrays are numpy, C-contiguous.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, "liboracle.so")
_REF_PATH = os.path.join(HERE, "_ref", "libref_bresenham.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-o", _LIB_PATH, src, "-lm"]
        try:  # OpenMP over the envs of update_occ_grid (the CPU baseline of bench.py uses every host core)
            subprocess.check_call(cmd[:2] + ["-fopenmp"] + cmd[2:], stderr=subprocess.DEVNULL)
        except subprocess.CalledProcessError:
            subprocess.check_call(cmd)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_post_process_depth.argtypes = [_f32p, _f32p, C.c_int64, C.c_float, _f32p, _f32p]
        L.orc_rgb_to_gray64.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_back_projection.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _u8p]
        L.orc_points_to_idx.argtypes = [_f32p, _u8p, _f32p, _f32p, C.c_int, C.c_int64, C.c_int, _i32p]
        L.orc_pose_to_idx.argtypes = [_f32p, _f32p, _f32p, C.c_int, _i64p]
        L.orc_bresenham3d.argtypes = [_i32p, _i32p, C.c_int, C.c_int, _i32p, _i32p]
        L.orc_update_occ_grid.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_gae_sb3.argtypes = [_f32p, _f32p, _u8p, _f32p, _u8p, C.c_int, C.c_int, C.c_double, C.c_double,
                                  _f32p, _f32p]
        L.orc_gae_rsl.argtypes = [_f32p, _f32p, _u8p, _f32p, C.c_int, C.c_int, C.c_double, C.c_double, _f32p, _f32p]
        for f in ("orc_post_process_depth", "orc_rgb_to_gray64", "orc_back_projection", "orc_points_to_idx",
                  "orc_pose_to_idx", "orc_bresenham3d", "orc_update_occ_grid", "orc_gae_sb3", "orc_gae_rsl"):
            getattr(L, f).restype = None
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def post_process_depth(depth_raw, seg_raw, depth_sense_dist=-50.0):
    d, s = _c(depth_raw, np.float32), _c(seg_raw, np.float32)
    do, so = np.empty_like(d), np.empty_like(s)
    lib().orc_post_process_depth(d.ravel(), s.ravel(), d.size, depth_sense_dist, do.ravel(), so.ravel())
    return do, so


def rgb_to_gray64(rgba, oh=64, ow=64):
    r = _c(rgba, np.uint8)
    n, h, w, _ = r.shape
    out = np.empty((n, 1, oh, ow), np.float32)
    lib().orc_rgb_to_gray64(r.ravel(), n, h, w, oh, ow, out.ravel())
    return out


def back_projection(depth, seg, c2w, inv_intri):
    d, s = _c(depth, np.float32), _c(seg, np.float32)
    n, h, w = d.shape
    world = np.empty((n, h * w, 3), np.float32)
    fg = np.empty((n, h * w), np.uint8)
    lib().orc_back_projection(d.ravel(), s.ravel(), _c(c2w, np.float32).ravel(), _c(inv_intri, np.float32).ravel(),
                              n, h, w, world.ravel(), fg.ravel())
    return world, fg.astype(bool)


def points_to_idx(world, fg, range_gt, voxel_size, g):
    wd = _c(world, np.float32)
    n, hw, _ = wd.shape
    idx = np.empty((n, hw, 3), np.int32)
    lib().orc_points_to_idx(wd.ravel(), _c(fg, np.uint8).ravel(), _c(range_gt, np.float32).ravel(),
                            _c(voxel_size, np.float32).ravel(), n, hw, g, idx.ravel())
    return idx


def pose_to_idx(poses_xyz, range_gt, voxel_size):
    p = _c(poses_xyz, np.float32)
    out = np.empty((p.shape[0], 3), np.int64)
    lib().orc_pose_to_idx(p.ravel(), _c(range_gt, np.float32).ravel(), _c(voxel_size, np.float32).ravel(),
                          p.shape[0], out.ravel())
    return out


def bresenham3d(source, targets, g):
    """Returns (trajectory_pts [R,3G,3] int32, lengths [R] int32) -- the raw kernel outputs."""
    t = _c(targets, np.int32).reshape(-1, 3)
    traj = np.zeros((t.shape[0], 3 * g, 3), np.int32)
    lens = np.zeros((t.shape[0],), np.int32)
    lib().orc_bresenham3d(_c(source, np.int32).ravel(), t.ravel(), t.shape[0], g, traj.ravel(), lens.ravel())
    return traj, lens


def update_occ_grid(depth, seg, c2w, inv_intri, poses_xyz, range_gt, voxel_size, grid_gt, prob_grid,
                    scanned_gt_grid, reset_mask=None, return_masks=False):
    """In-place on prob_grid / scanned_gt_grid (numpy f32 [N,G,G,G]); returns tri (+masks, coverage count)."""
    d, s = _c(depth, np.float32), _c(seg, np.float32)
    n, h, w = d.shape
    g = grid_gt.shape[1]
    assert prob_grid.dtype == np.float32 and prob_grid.flags.c_contiguous
    assert scanned_gt_grid.dtype == np.float32 and scanned_gt_grid.flags.c_contiguous
    tri = np.empty((n, g, g, g), np.float32)
    cov = np.zeros((n,), np.int32)
    hit = np.zeros((n, g, g, g), np.uint8) if return_masks else None
    path = np.zeros((n, g, g, g), np.uint8) if return_masks else None
    rm = None if reset_mask is None else _c(reset_mask, np.uint8)
    lib().orc_update_occ_grid(
        d.ravel(), s.ravel(), _c(c2w, np.float32).ravel(), _c(inv_intri, np.float32).ravel(),
        _c(poses_xyz, np.float32).ravel(), _c(range_gt, np.float32).ravel(), _c(voxel_size, np.float32).ravel(),
        _c(grid_gt, np.float32).ravel(), None if rm is None else rm.ctypes.data, n, h, w, g,
        prob_grid.ravel(), scanned_gt_grid.ravel(), tri.ravel(),
        None if hit is None else hit.ctypes.data, None if path is None else path.ctypes.data, cov.ctypes.data)
    if return_masks:
        return tri, cov, hit.astype(bool), path.astype(bool)
    return tri, cov


def gae_sb3(rewards, values, episode_starts, last_values, dones, gamma=0.99, gae_lambda=0.95):
    r = _c(rewards, np.float32)
    t, n = r.shape[:2]
    adv, ret = np.empty((t, n), np.float32), np.empty((t, n), np.float32)
    lib().orc_gae_sb3(r.ravel(), _c(values, np.float32).ravel(), _c(episode_starts, np.uint8).ravel(),
                      _c(last_values, np.float32).ravel(), _c(dones, np.uint8).ravel(), t, n, gamma, gae_lambda,
                      adv.ravel(), ret.ravel())
    return adv, ret


def gae_rsl(rewards, values, dones, last_values, gamma=0.99, lam=0.95):
    r = _c(rewards, np.float32)
    t, n = r.shape[:2]
    ret, adv = np.empty((t, n), np.float32), np.empty((t, n), np.float32)
    lib().orc_gae_rsl(r.ravel(), _c(values, np.float32).ravel(), _c(dones, np.uint8).ravel(),
                      _c(last_values, np.float32).ravel(), t, n, gamma, lam, ret.ravel(), adv.ravel())
    return ret, adv


def chamfer_distance_ref(x, y, chunk=2048):
    """TEST ORACLE, parity UNPINNED against the third-party library: pytorch3d (0.7.8, the version the reference's
    README names) is not in this image.  Restates its documented definition of
    `pytorch3d.loss.chamfer_distance(x[None], y[None])[0]` with default arguments (point_reduction="mean",
    batch_reduction="mean", norm=2, both directions): mean_i min_j |x_i-y_j|^2 + mean_j min_i |x_i-y_j|^2,
    as the reference uses it at gennbv/env/env_eval_gennbv.py:260-261.  float64 brute force."""
    import numpy as np
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)

    def one_way(a, b):
        out = np.empty(a.shape[0])
        for s0 in range(0, a.shape[0], chunk):
            d = a[s0:s0 + chunk, None, :] - b[None, :, :]
            out[s0:s0 + chunk] = (d * d).sum(-1).min(1)
        return out.mean()
    return one_way(x, y) + one_way(y, x)


def auc_update_ref(auc_rews, cur_rewards, cur_length, dones, episode_done_flag):
    """AUC_update of the reference, env by env (stable_baselines3/common/evaluation.py:358-378)."""
    import numpy as np
    a = np.array(auc_rews, dtype=np.float32, copy=True)
    for e in range(a.shape[0]):
        if episode_done_flag[e]:
            a[e, cur_length - 1] = a[e, cur_length - 2]
        elif dones[e] == 0:
            a[e, cur_length - 1] = cur_rewards[e]
    return a


# --------------------------------------------------------------------------- #
# the REFERENCE's own kernel text compiled as host C++ (oracle/build_ref.py)
# --------------------------------------------------------------------------- #
_ref = None


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref_bresenham3d(source, targets, g):
    """Runs the reference's ray_casting_kernel_3d (gennbv/utils.py:170-196) on the host."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF_PATH)
        _ref.ref_ray_casting_3d.argtypes = [_i32p, _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int]
        _ref.ref_ray_casting_3d.restype = None
    t = _c(targets, np.int32).reshape(-1, 3)
    traj = np.zeros((t.shape[0], 3 * g, 3), np.int32)
    lens = np.zeros((t.shape[0],), np.int32)
    _ref.ref_ray_casting_3d(_c(source, np.int32).ravel(), t.ravel(), traj.ravel(), lens.ravel(), t.shape[0], g, 3 * g)
    return traj, lens
