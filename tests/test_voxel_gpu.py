"""GPU: HIP state-encoding kernels (through the C-ABI) vs the CPU oracle and the goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests import golden_util as gu
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def test_bresenham_bit_exact_vs_golden_and_oracle():
    from gennbv_amd import utils
    fx = gu.load("F4_bresenham")
    for k in sorted(k[:-4] for k in fx.files if k.endswith("_src")):
        g = int(k.split("_")[0][1:])
        traj, lens = utils.bresenham3D_raw(T(fx[k + "_src"]), T(fx[k + "_tgt"]), g)
        assert np.array_equal(lens.cpu().numpy(), fx[k + "_len"]), k
        assert np.array_equal(traj.cpu().numpy(), fx[k + "_traj"].astype(np.int32)), k
    rs = np.random.RandomState(1)
    for g in (8, 20, 64, 128):
        src = rs.randint(-g, 2 * g, size=(1, 3)).astype(np.int32)
        tgt = rs.randint(0, g, size=(5000, 3)).astype(np.int32)
        otraj, olens = orc.bresenham3d(src, tgt, g)
        out = utils.bresenham3D_pycuda(T(src), T(tgt), g).cpu().numpy()
        ref = otraj[np.arange(3 * g)[None, :] < olens[:, None]].reshape(-1, 3)
        assert out.dtype == np.int64 and np.array_equal(out, ref)
    # empty target list: reference returns an empty [0,3] long tensor
    assert utils.bresenham3D_pycuda(T(src), torch.zeros(0, 3, device=DEV), 16).shape == (0, 3)


def test_postprocess_backprojection_voxelidx_vs_golden():
    from gennbv_amd import utils
    fx = gu.load("F2_backproj")
    n, g = int(fx["n"]), int(fx["g"])
    d, seg = gu.frames(fx)
    dp, sp = utils.post_process_depth(T(d[0]), T(seg[0]))
    assert dp.cpu().numpy().tobytes() == fx["depth_processed"].tobytes()
    assert sp.cpu().numpy().tobytes() == fx["seg_processed"].tobytes()
    c2w = T(fx["c2w"])
    pts, world, fg = utils.back_projection_fg(dp, sp, c2w, torch.from_numpy(fx["inv_intri"]), return_all=True)
    idx_lists = utils.scanned_pts_to_idx_3D(pts, T(fx["range_gt"]), T(fx["voxel_size"]), map_size=g)
    for e in range(n):
        assert pts[e].cpu().numpy().tobytes() == fx[f"world_{e}"].tobytes()
        assert np.array_equal(idx_lists[e].cpu().numpy(), fx[f"uidx_{e}"].reshape(-1, 3))
    assert np.array_equal(utils.pose_coord_to_idx_3D(T(fx["poses"][:, :3]), T(fx["range_gt"]), T(fx["voxel_size"]), g).cpu().numpy(), fx["pose_idx"])
    assert np.array_equal(utils.pose_coord_to_idx_3D(T(fx["far_poses"]), T(fx["range_gt"]), T(fx["voxel_size"]), g).cpu().numpy(), fx["far_idx"])
    # empty env list entry
    assert utils.scanned_pts_to_idx_3D([torch.zeros(0, 3, device=DEV)], T(fx["range_gt"]), T(fx["voxel_size"]), g) == [[]]


def _run_sequence(n, h, w, g, steps, seed, reset_at=(), pose_override=None, packed=None, gt_scale=None, max_steps=None,
                  int8_only=False, blank_envs=()):
    """HIP updater vs oracle on the same seeded synthetic frames; returns per-step mismatch info."""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=seed)
    frames = S.make_frames(scene, cfg, min(steps, 4), seed=seed, with_rgba=False)
    for k, e in enumerate(blank_envs):  # envs that see only background in some frames: empty ray lists (no walk task does any work)
        for f in frames[k % 2::2]:
            f.seg_raw[e] = 0.0
    kinv = S.inverse_intrinsics(h, w)
    if gt_scale is not None:
        scene.grid_gt = scene.grid_gt * gt_scale
    upd = OccupancyGridUpdater(n, g, h, w, kinv, scene.range_gt, scene.voxel_size, scene.grid_gt, DEV, packed=packed,
                               max_steps_between_resets=max_steps)
    assert upd.coded == (max_steps is not None)
    # the coded update clears its masks in passing (self_clean): every second sequence keeps them instead and compares them
    # with the oracle's; the self-cleaning ones are checked through the grids of the FOLLOWING steps (a left-over bit would
    # show up there)
    keep_masks = (seed % 2 == 0) or not upd.coded
    upd.self_clean = not keep_masks
    prob = np.zeros((n, g, g, g), np.float32)
    scan = np.zeros_like(prob)
    rs = np.random.RandomState(seed)
    for s in range(steps):
        f = frames[s % len(frames)]
        if pose_override is not None:
            f.poses = pose_override[s % len(pose_override)].clone()
        c2w = S.c2w_from_view(f.view, scene.env_origins)
        reset = None
        if s in reset_at:
            reset = (rs.rand(n) < 0.5).astype(np.uint8)
            reset[0] = 1
        dp, sp = orc.post_process_depth(f.depth_raw.numpy(), f.seg_raw.numpy())
        tri_o, cov_o, hit_o, path_o = orc.update_occ_grid(
            dp, sp, c2w.numpy(), kinv.numpy(), f.poses[:, :3].numpy(), scene.range_gt.numpy(), scene.voxel_size.numpy(),
            scene.grid_gt.numpy(), prob, scan, reset_mask=reset, return_masks=True)
        if int8_only:  # compact observations: the tri-class grid is written as int8 rows only
            t8 = torch.full((n, g ** 3), 99, dtype=torch.int8, device=DEV)
            assert upd.update(f.depth_raw.to(DEV), f.seg_raw.to(DEV), c2w.to(DEV), f.poses.to(DEV).contiguous(),
                              reset_mask=None if reset is None else torch.from_numpy(reset).to(DEV), tri_i8_out=t8, fp32_out=False) is None
            tri = t8.view(n, g, g, g).float()
        else:
            tri = upd.update(f.depth_raw.to(DEV), f.seg_raw.to(DEV), c2w.to(DEV), f.poses.to(DEV).contiguous(),
                             reset_mask=None if reset is None else torch.from_numpy(reset).to(DEV))
        if keep_masks:
            hit, path = upd.masks()
            assert np.array_equal(hit.cpu().numpy(), hit_o), f"step {s}: hit mask differs"
            assert np.array_equal(path.cpu().numpy(), path_o), f"step {s}: path mask differs"
        assert upd.prob_grid.cpu().numpy().tobytes() == prob.tobytes(), f"step {s}: prob grid differs"
        assert upd.scanned_gt_grid.cpu().numpy().tobytes() == scan.tobytes(), f"step {s}"
        assert tri.cpu().numpy().tobytes() == tri_o.tobytes(), f"step {s}"
        assert np.array_equal(upd.coverage_count.cpu().numpy(), cov_o), f"step {s}"
        assert hit_o.sum() > 0 and path_o.sum() > hit_o.sum()
    return upd


@pytest.mark.parametrize("n,h,w,g,steps", [(4, 240, 320, 16, 6), (3, 100, 100, 20, 5), (9, 120, 160, 64, 4),
                                           (2, 30, 37, 33, 3), (2, 60, 80, 96, 2), (2, 60, 80, 128, 2), (9, 48, 64, 128, 2)])
def test_fused_update_bit_exact_vs_oracle(n, h, w, g, steps):
    # (G = 96: 110 KiB LDS masks; G = 128: two 128 KiB windows per env; n = 9: more than one XCD group)
    _run_sequence(n, h, w, g, steps, seed=11 + g, reset_at=(2,))


@pytest.mark.parametrize("n,h,w,g,steps", [(4, 120, 160, 16, 14), (3, 100, 100, 20, 6), (5, 120, 160, 64, 5), (2, 30, 37, 33, 4)])
def test_coded_probability_grid_bit_exact_vs_oracle(n, h, w, g, steps):
    """1-byte coded prob grid (code = base << 7 | #path steps): decoded grid, tri-class grid, scanned set and coverage
    equal the oracle bit for bit over a sequence with resets (repeated -0.05 steps reach the fp32 values the reference
    reaches: -0.05, -0.1, -0.15000001, ...)."""
    _run_sequence(n, h, w, g, steps, seed=23 + g, reset_at=(2, 5), max_steps=100)


@pytest.mark.parametrize("n,h,w,g,steps", [(4, 120, 160, 16, 8), (3, 100, 100, 20, 6), (3, 120, 160, 64, 4), (2, 30, 37, 33, 4),
                                           (2, 30, 37, 18, 4)])
def test_coded_update_int8_only_rows_bit_exact_vs_oracle(n, h, w, g, steps):
    """Compact observation rows (no fp32 tri-class output): G^3 % 16 == 0 takes the 16-voxels-per-lane kernel,
    G = 18 the 4-per-lane one, G = 33 the scalar one."""
    _run_sequence(n, h, w, g, steps, seed=31 + g, reset_at=(2, 5), max_steps=100, int8_only=True)


@pytest.mark.parametrize("seed", [40, 41])  # even: masks kept and compared (memset per call); odd: self-cleaning workspace
def test_coded_update_uneven_load_67_envs(seed):
    """67 envs (more than eight XCD groups, not a multiple of 8), envs whose ray list is EMPTY in some frames, envs with one ray-list slice
    and envs with many, resets, six consecutive calls on one workspace -- hit / path masks, probability codes, scanned sets, coverage
    and int8 rows against the oracle at every step.  (Written for round 4's persistent ray-walk + grid-update launch, which was
    measured slower and removed; the uneven-load case stays.)"""
    _run_sequence(67, 60, 80, 64, 6, seed=seed, reset_at=(2, 4), max_steps=100, int8_only=True, blank_envs=(0, 3, 8, 17, 66))


def test_coded_probability_grid_tables_and_saturation():
    import ctypes as C
    from gennbv_amd import _lib
    lib = _lib.load()
    pl, tl = (C.c_float * 256)(), (C.c_float * 256)()
    lib.gnbv_prob_code_tables(pl, tl)
    x = np.float32(0.0)
    for k in range(128):
        assert np.float32(pl[k]) == x and tl[k] == float(x > 0.5) - float(x < 0.0)
        x = np.float32(x - np.float32(0.05))
    x = np.float32(1.0)
    for k in range(128):
        assert np.float32(pl[128 + k]) == x and tl[128 + k] == float(x > 0.5) - float(x < 0.0)
        x = np.float32(x - np.float32(0.05))
    # more path steps than the declared bound: the saturation flag turns reading the grid into an error
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    n, h, w, g = 2, 60, 80, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=3)
    f = S.make_frames(scene, cfg, 1, seed=3, with_rgba=False)[0]
    upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, DEV,
                               max_steps_between_resets=100)
    c2w = S.c2w_from_view(f.view, scene.env_origins).to(DEV)
    for _ in range(130):
        upd.update(f.depth_raw.to(DEV), f.seg_raw.to(DEV), c2w, f.poses.to(DEV).contiguous())
    with pytest.raises(_lib.GennbvHipError):
        upd.prob_grid


def test_coded_update_byte_parallel_tri_class_saturation_and_other_tables():
    """The 16-voxels-per-lane update classifies four codes per dword with integer arithmetic when the tri-class table has the shape
    gnbv_prob_code_tables gives it (+1 / 0 / -1 runs per base); ANY other table takes the table reads.  Both against the table applied
    to the codes on the host; the byte-parallel step saturates at 127 path steps and raises the overflow flag like the scalar one."""
    from gennbv_amd import _lib
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    n, h, w, g = 3, 60, 80, 32
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=5)
    frames = S.make_frames(scene, cfg, 3, seed=5, with_rgba=False)
    rng = np.random.default_rng(7)
    for table in ("default", "random", "shifted"):
        upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, DEV,
                                   max_steps_between_resets=100)
        if table == "random":  # not of the run shape: table reads
            upd._tri_lut = torch.from_numpy(rng.integers(-1, 2, 256).astype(np.float32)).to(DEV)
        elif table == "shifted":  # run shape with other thresholds (K1 = 0 for base 0, K2 = 128 for base 1)
            t = np.concatenate([np.r_[np.zeros(40), -np.ones(88)], np.r_[np.ones(3), np.zeros(125)]]).astype(np.float32)
            upd._tri_lut = torch.from_numpy(t).to(DEV)
        lut = upd._tri_lut.cpu().numpy()
        t8 = torch.full((n, g ** 3), 99, dtype=torch.int8, device=DEV)
        for s in range(8):
            f = frames[s % 3]
            reset = torch.tensor([0, 1, 0], dtype=torch.uint8, device=DEV) if s == 4 else None
            upd.update(f.depth_raw.to(DEV), f.seg_raw.to(DEV), S.c2w_from_view(f.view, scene.env_origins).to(DEV), f.poses.to(DEV).contiguous(),
                       reset_mask=reset, tri_i8_out=t8, fp32_out=False)
            code = upd.prob_code.cpu().numpy().reshape(n, -1)
            assert np.array_equal(t8.cpu().numpy(), lut[code].astype(np.int8)), (table, s)
        assert (code & 127).max() >= 3 and (code >= 128).any()
    # saturation: more path steps than a byte holds
    f = frames[0]
    c2w = S.c2w_from_view(f.view, scene.env_origins).to(DEV)
    for _ in range(130):
        upd.update(f.depth_raw.to(DEV), f.seg_raw.to(DEV), c2w, f.poses.to(DEV).contiguous(), tri_i8_out=t8, fp32_out=False)
    assert (upd.prob_code.cpu().numpy() & 127).max() == 127
    with pytest.raises(_lib.GennbvHipError):
        upd.prob_grid


def test_f32_path_and_non_binary_ground_truth():
    """packed=False keeps the reference's fp32 scanned/gt tensors; a non-binary GT (0.4 per voxel:
    scanned accumulates 0.4, 0.8, then clips at 1) automatically takes that path."""
    upd = _run_sequence(3, 60, 80, 16, 4, seed=8, reset_at=(2,), packed=False)
    assert not upd.packed
    upd = _run_sequence(3, 60, 80, 16, 5, seed=8, gt_scale=0.4)
    assert not upd.packed and float(upd.scanned_gt_grid.max()) == 1.0
    assert bool(((upd.scanned_gt_grid > 0) & (upd.scanned_gt_grid < 1)).any())
    upd = _run_sequence(3, 60, 80, 16, 2, seed=8)
    assert upd.packed


@pytest.mark.parametrize("n,h,w,g,steps,planes", [(4, 120, 160, 16, 6, "3"), (3, 100, 100, 20, 5, "2"), (5, 120, 160, 64, 4, "8"), (2, 30, 37, 33, 4, "32"),
                                                  (9, 48, 64, 128, 2, ""), (2, 60, 80, 128, 3, "48")])
def test_large_grid_kernels_bit_exact_vs_oracle(n, h, w, g, steps, planes, monkeypatch):
    """k_hit_atomic (hit bits by returning global atomics, the first setter lists the voxel) + k_ray_slab (every ray walked slab by
    slab of x-planes from the closed form of the reference's Bresenham): the path grids above G = 104 take by default, forced here
    at every size (GENNBV_VOXEL_LARGE=1) with slab heights that do not divide the grid (GENNBV_VOXEL_SLAB_PLANES; rounded up to the
    word alignment: G = 20 -> 2, G = 33 -> 32 planes); masks, coded grids, tri-class rows and coverage against the oracle over a
    sequence with resets."""
    monkeypatch.setenv("GENNBV_VOXEL_LARGE", "1")
    if planes:
        monkeypatch.setenv("GENNBV_VOXEL_SLAB_PLANES", planes)
    _run_sequence(n, h, w, g, steps, seed=40 + g, reset_at=(2,), max_steps=100)
    _run_sequence(n, h, w, g, min(steps, 3), seed=41 + g, reset_at=(1,), max_steps=100, int8_only=(g ** 3 % 4 == 0))


def test_large_grid_kernels_sources_outside_the_grid_and_round1_kernels(monkeypatch):
    """Ray sources outside the grid through the slab walk (bounds-tested steps), and the round-1 window kernels at 128^3
    (GENNBV_VOXEL_LARGE=0), which stay the path of a workspace without ray lists."""
    far = [torch.tensor([[100.0, -60.0, 40.0, 0, 0, 0], [-8.9, 8.9, -3.0, 0, 0, 0], [0.0, 0.0, 300.0, 0, 0, 0],
                         [2000.0, 900.0, 5000.0, 0, 0, 0]]),
           torch.tensor([[-700.0, 0.3, 0.2, 0, 0, 0], [8.0, 8.0, 10.0, 0, 0, 0], [-8.0, -8.0, 0.0, 0, 0, 0],
                         [0.1, 0.1, 0.1, 0, 0, 0]])]
    monkeypatch.setenv("GENNBV_VOXEL_LARGE", "1")
    for g, planes in ((16, "5"), (64, "16")):
        monkeypatch.setenv("GENNBV_VOXEL_SLAB_PLANES", planes)
        _run_sequence(4, 60, 80, g, 2, seed=5, pose_override=far)
    monkeypatch.delenv("GENNBV_VOXEL_SLAB_PLANES")
    monkeypatch.setenv("GENNBV_VOXEL_LARGE", "0")
    _run_sequence(2, 60, 80, 128, 2, seed=139, reset_at=(1,))


def test_ray_source_far_outside_grid():
    """Ray sources the lattice never produces: far outside the grid (closed form with large
    deltas, and the sequential fallback beyond kMaxClosedFormDelta), below / beside the grid."""
    n = 4
    far = [torch.tensor([[100.0, -60.0, 40.0, 0, 0, 0], [-8.9, 8.9, -3.0, 0, 0, 0], [0.0, 0.0, 300.0, 0, 0, 0],
                         [2000.0, 900.0, 5000.0, 0, 0, 0]]),
           torch.tensor([[-700.0, 0.3, 0.2, 0, 0, 0], [8.0, 8.0, 10.0, 0, 0, 0], [-8.0, -8.0, 0.0, 0, 0, 0],
                         [0.1, 0.1, 0.1, 0, 0, 0]])]
    for g in (16, 64):
        _run_sequence(n, 60, 80, g, 2, seed=5, pose_override=far)


def test_fused_update_matches_reference_golden_masks():
    """Golden F5 (reference env on CPU): prob/tri/scanned after real reference steps."""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    fx = gu.load("F5_envstep_g20")
    n, h, w, g = int(fx["n"]), int(fx["h"]), int(fx["w"]), int(fx["g"])
    d, seg = gu.frames(fx)
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    gt = gu.unpack_bits(fx["grid_gt_bits"], g)
    upd = OccupancyGridUpdater(n, g, h, w, torch.from_numpy(fx["inv_intri"]), torch.from_numpy(fx["range_gt"]),
                               torch.from_numpy(fx["voxel_size"]), torch.from_numpy(gt), DEV)
    org = torch.from_numpy(fx["env_origins"])
    # reset(): frame 0 at the init pose
    init_pose = torch.tensor(cfg.init_pose_buf).repeat(n, 1).to(DEV)
    c2w = S.c2w_from_view(torch.from_numpy(fx["view"][0]), org).to(DEV)
    tri = upd.update(T(d[0]), T(seg[0]), c2w, init_pose)
    assert np.array_equal(tri.cpu().numpy().astype(np.int8), fx["reset_tri"])
    assert upd.prob_grid.cpu().numpy().tobytes() == fx["reset_prob"].tobytes()


def test_full_size_properties_config1():
    """BASELINE config 1 size (256 envs, 240x320, 64^3): size-independent properties."""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    n, h, w, g = 256, 240, 320, 64
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=3, device=DEV)
    f = S.make_frames(scene, cfg, 1, seed=3, with_rgba=False)[0]
    upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, DEV)
    upd.self_clean = False  # (the masks are inspected)
    c2w = S.c2w_from_view(f.view, scene.env_origins)
    tri1 = upd.update(f.depth_raw, f.seg_raw, c2w, f.poses.contiguous()).clone()
    hit, path = upd.masks()
    prob1 = upd.prob_grid.clone()
    # hit voxels are occupied, path-only voxels are free, everything else unknown
    assert bool((prob1[hit] == 1.0).all())
    assert bool((prob1[path & ~hit] == -0.05).all())
    assert bool((prob1[~path & ~hit] == 0.0).all())
    assert bool(hit.flatten(1).any(1).float().mean() > 0.9)
    # every hit voxel is the endpoint of its own ray -> hit subset of path (source->target, endpoint included)
    assert bool((path | ~hit).all())
    assert bool((tri1 == (prob1 > 0.5).float() - (prob1 < 0).float()).all())
    # coverage count == sum(scanned) and scanned == hit & gt for binary gt after one step
    assert torch.equal(upd.coverage_count.long(), upd.scanned_gt_grid.flatten(1).sum(1).long())
    assert torch.equal(upd.scanned_gt_grid, (hit & (scene.grid_gt > 0)).float())
    # idempotence of the sets: same frame again leaves hit voxels at 1, decrements path-only voxels once more
    upd.update(f.depth_raw, f.seg_raw, c2w, f.poses.contiguous())
    hit2, path2 = upd.masks()
    assert torch.equal(hit, hit2) and torch.equal(path, path2)
    assert bool((upd.prob_grid[path & ~hit] == torch.tensor(-0.05, device=DEV) - 0.05).all())
    # a reset mask zeroes history: result equals the first step again
    upd.update(f.depth_raw, f.seg_raw, c2w, f.poses.contiguous(), reset_mask=torch.ones(n, dtype=torch.uint8, device=DEV))
    assert torch.equal(upd.prob_grid, prob1)
    # spot-check 3 envs bit-exactly against the oracle at full resolution
    sel = [0, 101, 255]
    pr = np.zeros((3, g, g, g), np.float32); sc = np.zeros_like(pr)
    dp, sp = orc.post_process_depth(f.depth_raw[sel].cpu().numpy(), f.seg_raw[sel].cpu().numpy())
    tri_o, cov_o = orc.update_occ_grid(dp, sp, c2w[sel].cpu().numpy(), S.inverse_intrinsics(h, w).numpy(), f.poses[sel, :3].cpu().numpy(),
                                       scene.range_gt[sel].cpu().numpy(), scene.voxel_size[sel].cpu().numpy(), scene.grid_gt[sel].cpu().numpy(), pr, sc)
    assert prob1[sel].cpu().numpy().tobytes() == pr.tobytes()
    assert tri1[sel].cpu().numpy().tobytes() == tri_o.tobytes()


@pytest.mark.parametrize("large", ["0", "1"])  # "1": k_hit_atomic (the kernel of grids above 104^3) forced at 64^3
def test_hit_mask_of_all_256_full_size_envs_equals_the_per_pixel_oracle_voxels(large, monkeypatch):
    """Every voxel a foreground pixel lands in -- per pixel, by the oracle's canonical chain (A1 post-processing, A2 back-projection,
    A3 index; gennbv/utils.py:230-270, env_train_gennbv.py:277-299) -- must be in the hit mask, and nothing else: ALL 256 envs of
    BASELINE configs[1]'s geometry, not a sample of three.  (Round 3's tools/debug_hit_mask.py as a test: it is the check that caught a
    k_hit_list variant losing 6 single-pixel voxels in 256 envs -- voxels one pixel wide are exactly what a sampled check misses.)"""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    monkeypatch.setenv("GENNBV_VOXEL_LARGE", large)
    n, h, w, g = 256, 240, 320, 64
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=1, device=DEV)
    kinv = S.inverse_intrinsics(h, w)
    upd = OccupancyGridUpdater(n, g, h, w, kinv, scene.range_gt, scene.voxel_size, scene.grid_gt, DEV)
    upd.self_clean = False
    orc.lib().orc_set_num_threads(min(64, os.cpu_count() or 1))
    total_single = 0
    for seed in (1, 2):  # two poses per env
        f = S.make_frames(scene, cfg, 1, seed=seed, with_rgba=False)[0]
        c2w = S.c2w_from_view(f.view, scene.env_origins)
        upd.update(f.depth_raw, f.seg_raw, c2w, f.poses.contiguous(), reset_mask=torch.ones(n, dtype=torch.uint8, device=DEV))
        hit, _ = upd.masks()
        hit = hit.cpu().numpy().reshape(n, -1)
        dp, sp = orc.post_process_depth(f.depth_raw.cpu().numpy(), f.seg_raw.cpu().numpy())
        world, fg = orc.back_projection(dp, sp, c2w.cpu().numpy(), kinv.numpy())
        idx = orc.points_to_idx(world, fg, scene.range_gt.cpu().numpy(), scene.voxel_size.cpu().numpy(), g).reshape(n, h * w, 3)
        keep = idx[..., 0] >= 0  # (background / out-of-range pixels: -1)
        lin = (idx[..., 0].astype(np.int64) * g + idx[..., 1]) * g + idx[..., 2]
        for e in range(n):
            cnt = np.bincount(lin[e][keep[e]], minlength=g ** 3)
            ref = cnt > 0
            total_single += int((cnt == 1).sum())
            if not np.array_equal(ref, hit[e]):
                miss, extra = np.nonzero(ref & ~hit[e])[0], np.nonzero(~ref & hit[e])[0]
                raise AssertionError(f"env {e} (frame seed {seed}): {len(miss)} voxels missing {miss[:6]}, {len(extra)} extra {extra[:6]}; "
                                     f"pixels of the first missing voxel: {np.nonzero(keep[e] & (lin[e] == (miss[0] if len(miss) else -1)))[0][:8]}")
    assert total_single > 1000  # the case the sampled checks cannot see is present: voxels hit by exactly one pixel


def test_tri_written_into_strided_observation_rows():
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    n, h, w, g = 5, 48, 64, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=9, device=DEV)
    f = S.make_frames(scene, cfg, 1, seed=9, with_rgba=False)[0]
    upd = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, DEV)
    obs = torch.full((n, cfg.obs_dim), 7.0, device=DEV)
    c2w = S.c2w_from_view(f.view, scene.env_origins)
    upd.update(f.depth_raw, f.seg_raw, c2w, f.poses.contiguous(), tri_out=obs[:, cfg.state_dim:], tri_row_stride=cfg.obs_dim)
    ref = OccupancyGridUpdater(n, g, h, w, S.inverse_intrinsics(h, w), scene.range_gt, scene.voxel_size, scene.grid_gt, DEV)
    tri = ref.update(f.depth_raw, f.seg_raw, c2w, f.poses.contiguous())
    assert torch.equal(obs[:, cfg.state_dim:cfg.state_dim + g ** 3], tri.view(n, -1))
    assert bool((obs[:, :cfg.state_dim] == 7).all()) and bool((obs[:, cfg.state_dim + g ** 3:] == 7).all())


def test_bad_arguments_return_error_codes():
    from gennbv_amd import _lib
    lib = _lib.load()
    assert lib.gnbv_update_occ_grid(*([None] * 5), 3, *([None] * 4), 1, 1, 1, 2, -50.0, None, None, None, 8, None, None, 0, None) == 1
    assert lib.gnbv_gae_sb3(None, None, None, None, None, 1, 1, 0.99, 0.95, None, None, None) == 1


@pytest.mark.parametrize("large", ["", "1"])
def test_non_pinhole_intrinsics_take_the_generic_path(large, monkeypatch):
    """inv_intri with non-zero skew terms disables the exact-zero shortcut (k_hit_list<false> / k_hit_mask<false>; large = "1":
    k_hit_atomic<false>, the canonical chain over every pixel through the atomics)."""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    if large:
        monkeypatch.setenv("GENNBV_VOXEL_LARGE", large)
    n, h, w, g = 3, 60, 80, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=21)
    f = S.make_frames(scene, cfg, 1, seed=21, with_rgba=False)[0]
    kinv = S.inverse_intrinsics(h, w).clone()
    kinv[0, 1] = 3e-4; kinv[1, 0] = -2e-4; kinv[2, 0] = 1e-6; kinv[2, 2] = 1.0009765625
    c2w = S.c2w_from_view(f.view, scene.env_origins)
    upd = OccupancyGridUpdater(n, g, h, w, kinv, scene.range_gt, scene.voxel_size, scene.grid_gt, DEV)
    tri = upd.update(f.depth_raw.to(DEV), f.seg_raw.to(DEV), c2w.to(DEV), f.poses.to(DEV).contiguous())
    prob = np.zeros((n, g, g, g), np.float32); scan = np.zeros_like(prob)
    dp, sp = orc.post_process_depth(f.depth_raw.numpy(), f.seg_raw.numpy())
    tri_o, cov_o = orc.update_occ_grid(dp, sp, c2w.numpy(), kinv.numpy(), f.poses[:, :3].numpy(), scene.range_gt.numpy(),
                                       scene.voxel_size.numpy(), scene.grid_gt.numpy(), prob, scan)
    assert tri.cpu().numpy().tobytes() == tri_o.tobytes()
    assert upd.prob_grid.cpu().numpy().tobytes() == prob.tobytes()


@pytest.mark.parametrize("large", ["", "1"])
def test_special_depth_values_in_fused_path(large, monkeypatch):
    """NaN / +-inf / < -50 raw depths and NaN seg inside the fused kernel (A1 fused; large = "1": inside k_hit_atomic)."""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    if large:
        monkeypatch.setenv("GENNBV_VOXEL_LARGE", large)
    n, h, w, g = 2, 48, 64, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=4)
    f = S.make_frames(scene, cfg, 1, seed=4, with_rgba=False)[0]
    rs = np.random.RandomState(0)
    d = f.depth_raw.clone().view(-1); sg = f.seg_raw.clone().view(-1)
    idx = torch.from_numpy(rs.choice(d.numel(), 600, replace=False))
    vals = torch.tensor([float("nan"), float("inf"), -float("inf"), -75.5, -50.0, -49.99, 0.0, -0.0, 3.0e38, -3.0e38] * 60)
    d[idx] = vals
    sg[idx[:300]] = 255.0  # make half of them foreground so they reach the math
    sg[idx[300:330]] = float("nan"); sg[idx[330:360]] = float("inf"); sg[idx[360:390]] = -float("inf")
    d, sg = d.view(n, h, w), sg.view(n, h, w)
    kinv = S.inverse_intrinsics(h, w)
    c2w = S.c2w_from_view(f.view, scene.env_origins)
    upd = OccupancyGridUpdater(n, g, h, w, kinv, scene.range_gt, scene.voxel_size, scene.grid_gt, DEV)
    tri = upd.update(d.to(DEV), sg.to(DEV), c2w.to(DEV), f.poses.to(DEV).contiguous())
    prob = np.zeros((n, g, g, g), np.float32); scan = np.zeros_like(prob)
    dp, sp = orc.post_process_depth(d.numpy(), sg.numpy())
    tri_o, _ = orc.update_occ_grid(dp, sp, c2w.numpy(), kinv.numpy(), f.poses[:, :3].numpy(), scene.range_gt.numpy(),
                                   scene.voxel_size.numpy(), scene.grid_gt.numpy(), prob, scan)
    assert tri.cpu().numpy().tobytes() == tri_o.tobytes()
    assert upd.prob_grid.cpu().numpy().tobytes() == prob.tobytes()


@pytest.mark.parametrize("n,m", [(5000, 3000), (1, 1), (1025, 7), (3, 4097)])
def test_chamfer_distance_vs_float64_brute_force(n, m):
    """gnbv_chamfer_distance (eval accuracy, env_eval_gennbv.py:253-262 / pytorch3d definition) against the float64
    oracle; tolerance = fp32 rounding of the squared distances (difference form, no cancellation)."""
    from gennbv_amd.eval import metrics as M
    from oracle import oracle
    gen = torch.Generator().manual_seed(n * 31 + m)
    x = (torch.rand(n, 3, generator=gen) - 0.5) * 16.0
    y = x[torch.randint(0, n, (m,), generator=gen)] + 0.01 * torch.randn(m, 3, generator=gen) if n > 1 else torch.rand(m, 3, generator=gen)
    ref = oracle.chamfer_distance_ref(x.numpy(), y.numpy())
    got = float(M.chamfer_distance(x.to("cuda:0"), y.to("cuda:0")))
    assert abs(got - ref) <= 2e-6 * max(ref, 1e-12) + 1e-12, (got, ref)
    assert float(M.chamfer_distance(x.to("cuda:0"), x.to("cuda:0"))) == 0.0
    # the accuracy metric: 1 cm rounding + unique + chamfer, x 100
    acc = float(M.reconstruction_accuracy_cm(x.to("cuda:0"), y.to("cuda:0")))
    xr = np.unique(np.round(x.numpy().astype(np.float32) * np.float32(100.0)) / np.float32(100.0), axis=0)
    assert abs(acc - 100.0 * oracle.chamfer_distance_ref(xr, y.numpy())) <= 1e-4 * max(acc, 1e-9) + 1e-9


@pytest.mark.parametrize("large", ["", "1"])
def test_predictor_queue_overflow_falls_back_to_the_canonical_chain(large, monkeypatch):
    """k_hit_list decides a pixel with the voxel-space predictor only when its quotients keep clear of every voxel boundary;
    the rest is queued for the canonical chain, and a queue that overflows (2048 per workgroup) re-runs the whole chunk.
    A camera looking straight down at a floor that lies EXACTLY on a voxel boundary plane puts every pixel of env 0 within
    round-off of an integer quotient; env 1 sees the same floor half a voxel higher (all pixels decided by the predictor) with a
    band of special depths (queued, no overflow).  Both must equal the oracle bit for bit."""
    from gennbv_amd.env.state_encoding import OccupancyGridUpdater
    if large:  # the same stream / queue / fallback inside k_hit_atomic
        monkeypatch.setenv("GENNBV_VOXEL_LARGE", large)
    n, h, w, g = 2, 240, 320, 16
    cfg = TaskConfig(camera_width=w, camera_height=h, grid_size=g)
    scene = S.make_scenes(n, g, seed=11)
    kinv = S.inverse_intrinsics(h, w)
    vox_z = float(scene.voxel_size[0, 2])
    vmin_z = float(np.float32(scene.range_gt[0, 5]) - np.float32(0.5) * np.float32(vox_z))
    cam_z = 9.0
    c2w = torch.zeros(n, 4, 4)
    c2w[:, 0, 0], c2w[:, 1, 1], c2w[:, 2, 2], c2w[:, 3, 3] = 1.0, -1.0, -1.0, 1.0
    c2w[:, 2, 3] = cam_z
    floor_z = [vmin_z + 3.0 * vox_z, vmin_z + 3.5 * vox_z]
    depth = torch.empty(n, h, w)
    for e in range(n):
        depth[e] = -(cam_z - floor_z[e])  # Isaac convention: negative metres
    depth[1, 100:104] = torch.tensor([float("nan"), -float("inf"), -80.0, float("inf")]).view(4, 1)
    seg = torch.full((n, h, w), 255.0)
    poses = torch.tensor([[0.0, 0.0, cam_z, 0, 0, 0]] * n)
    upd = OccupancyGridUpdater(n, g, h, w, kinv, scene.range_gt, scene.voxel_size, scene.grid_gt, DEV)
    tri = upd.update(depth.to(DEV), seg.to(DEV), c2w.to(DEV), poses.to(DEV).contiguous())
    prob = np.zeros((n, g, g, g), np.float32); scan = np.zeros_like(prob)
    dp, sp = orc.post_process_depth(depth.numpy(), seg.numpy())
    tri_o, cov_o = orc.update_occ_grid(dp, sp, c2w.numpy(), kinv.numpy(), poses[:, :3].numpy(), scene.range_gt.numpy(),
                                       scene.voxel_size.numpy(), scene.grid_gt.numpy(), prob, scan)
    assert (prob == 1.0).sum() > 50  # the floor is inside the grid
    assert tri.cpu().numpy().tobytes() == tri_o.tobytes()
    assert upd.prob_grid.cpu().numpy().tobytes() == prob.tobytes()
    assert np.array_equal(upd.coverage_count.cpu().numpy(), cov_o)
