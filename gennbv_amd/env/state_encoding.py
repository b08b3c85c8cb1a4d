"""Fused state encoding: counterpart of Env_Train_GenNBV.update_occ_grid
(gennbv/env/env_train_gennbv.py:277-326) + post_process_camera_tensor's depth/seg
branch (gennbv/env/env_train_base.py:521-534) + back_projection_fg (:494-533).

One `update()` call = one environment step for all N envs = three gfx950 kernel
launches (gennbv_amd/csrc/voxel.hip) instead of the reference's N-iteration
Python loop.  State tensors keep the reference's names, shapes and dtypes:

    prob_grid, scanned_gt_grid, grid_gt : [N,G,G,G] f32
    occ_grids_tri_cls                   : [N,G,G,G] f32 view (may alias a slice of
                                          the flat observation rows)
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from .. import _lib


class OccupancyGridUpdater:
    def __init__(self, num_envs: int, grid_size: int, camera_height: int, camera_width: int,
                 inv_intri: torch.Tensor, range_gt: torch.Tensor, voxel_size_gt: torch.Tensor, grid_gt: torch.Tensor,
                 device, depth_sense_dist: float = -50.0, packed: Optional[bool] = None,
                 max_steps_between_resets: Optional[int] = None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.GennbvHipError("OccupancyGridUpdater needs a GPU device (no CPU fallback)")
        self.num_envs, self.grid_size = int(num_envs), int(grid_size)
        self.h, self.w = int(camera_height), int(camera_width)
        self.depth_sense_dist = float(depth_sense_dist)
        g = self.grid_size
        self.inv_intri_host = inv_intri.detach().to("cpu", torch.float32).contiguous()
        self.range_gt = range_gt.to(self.device, torch.float32).contiguous()
        self.voxel_size_gt = voxel_size_gt.to(self.device, torch.float32).contiguous()
        self.grid_gt = grid_gt.to(self.device, torch.float32).contiguous()
        assert self.grid_gt.shape == (num_envs, g, g, g)
        self.coded = False
        self._prob_f32 = torch.zeros(num_envs, g, g, g, dtype=torch.float32, device=self.device)
        # Binary ground truth (the reference's GT is an occupancy indicator): keep grid_gt and the
        # scanned set as bitmasks -- identical results, half the HBM traffic of the streaming pass.
        words = self.lib.gnbv_grid_bit_words(g)
        self.gt_bits = torch.zeros(num_envs, words, dtype=torch.int32, device=self.device)
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.gnbv_pack_grid_bits(self.grid_gt.data_ptr(), num_envs, g, self.gt_bits.data_ptr(), flag.data_ptr(),
                                                _lib.stream_ptr(self.device)), "gnbv_pack_grid_bits")
        binary = int(flag.item()) == 0
        self.packed = binary if packed is None else (bool(packed) and binary)
        self.scanned_bits = torch.zeros(num_envs, words, dtype=torch.int32, device=self.device)
        self._scanned_f32 = None if self.packed else torch.zeros_like(self._prob_f32)
        self.coverage_count = torch.zeros(num_envs, dtype=torch.int32, device=self.device)
        nbytes = self.lib.gnbv_voxel_workspace_bytes_hw(num_envs, g, self.h, self.w)  # masks + per-env ray lists
        # torch's caching allocator returns >=512-byte aligned blocks
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        # coded update: the grid-update launch clears the masks / ray counts it consumed, so no fill launch per step
        # (GNBV_VOXEL_WS_CLEAN).  Switch off BEFORE the first update to keep the masks of the last step for `masks()`.
        self.self_clean = True
        self._ws_dirty = False
        # measurement hook (bench.py): an object with `start()` / `stop()` that records HIP events on the current stream IMMEDIATELY around
        # the library call -- the launch's duration without the Python in front of it (argument checks, tensor bookkeeping: ~10-15 us that
        # an event pair around update() counts as kernel time whenever the GPU is ahead of the host)
        self.timing = None
        assert self.workspace.data_ptr() % 256 == 0
        self._own_tri = None
        # Coded probability grid (1 byte per voxel, exact): only when the GT is binary (packed mode) and the caller
        # guarantees that no voxel sees more than 127 path steps between two resets (episode length <= 127).
        self.coded = bool(self.packed and max_steps_between_resets is not None and 0 < int(max_steps_between_resets) <= 127
                          )
        if self.coded:
            import ctypes as C
            pl, tl = (C.c_float * 256)(), (C.c_float * 256)()
            self.lib.gnbv_prob_code_tables(pl, tl)
            self._prob_lut = torch.tensor(list(pl), dtype=torch.float32, device=self.device)
            self._tri_lut = torch.tensor(list(tl), dtype=torch.float32, device=self.device)
            self.prob_code = torch.zeros(num_envs, g ** 3, dtype=torch.uint8, device=self.device)
            self.code_overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._prob_f32 = None

    @property
    def prob_grid(self) -> torch.Tensor:
        """The reference's fp32 [N,G,G,G] probability grid (env_train_gennbv.py:180-181); in coded mode it is
        decoded from the byte codes on demand (and the saturation flag is checked)."""
        if not self.coded:
            return self._prob_f32
        if int(self.code_overflow.item()) != 0:
            raise _lib.GennbvHipError("coded probability grid: a voxel saw more than 127 path steps between resets "
                                      "(pass a correct max_steps_between_resets, or none: the fp32 probability grid is used then)")
        n, g = self.num_envs, self.grid_size
        out = torch.empty(n, g, g, g, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.gnbv_decode_prob_grid(self.prob_code.data_ptr(), out.numel(), self._prob_lut.data_ptr(), out.data_ptr(),
                                                  _lib.stream_ptr(self.device)), "gnbv_decode_prob_grid")
        return out

    def update(self, depth_raw: torch.Tensor, seg_raw: torch.Tensor, c2w: torch.Tensor, poses: torch.Tensor,
               reset_mask: Optional[torch.Tensor] = None, tri_out: Optional[torch.Tensor] = None,
               tri_row_stride: Optional[int] = None, tri_i8_out: Optional[torch.Tensor] = None,
               fp32_out: bool = True) -> Optional[torch.Tensor]:
        """depth_raw/seg_raw [N,H,W] f32 RAW camera tensors, c2w [N,4,4] f32, poses [N,>=3]
        f32 (xyz first).  `tri_out`: optional destination whose row e starts at
        tri_out.data_ptr() + e*tri_row_stride*4 (e.g. the grid slice of the flat obs).
        `fp32_out=False` (coded mode with `tri_i8_out`): compact observations, the tri-class grid is written
        only as int8."""
        n, g = self.num_envs, self.grid_size
        _lib.require_cuda(depth_raw, seg_raw, c2w, poses, reset_mask, tri_out)
        _lib.require_contig(depth_raw, seg_raw, c2w)
        assert depth_raw.shape == (n, self.h, self.w) and depth_raw.dtype == torch.float32
        assert seg_raw.shape == (n, self.h, self.w) and seg_raw.dtype == torch.float32
        assert c2w.shape == (n, 4, 4) and c2w.dtype == torch.float32
        assert poses.dtype == torch.float32 and poses.stride(-1) == 1 and poses.shape[0] == n
        if not fp32_out:
            assert self.coded and tri_i8_out is not None and tri_out is None, "int8-only output needs the coded update and tri_i8_out"
        elif tri_out is None:
            if self._own_tri is None:
                self._own_tri = torch.empty(n, g, g, g, dtype=torch.float32, device=self.device)
            tri_out, tri_row_stride = self._own_tri, g ** 3
        if reset_mask is not None:
            assert reset_mask.dtype == torch.uint8 and reset_mask.is_contiguous()
        if tri_i8_out is not None:
            assert self.coded, "the int8 copy of the tri-class grid is produced by the coded update only"
            assert tri_i8_out.dtype == torch.int8 and tri_i8_out.shape == (n, g ** 3) and tri_i8_out.stride(1) == 1
        if self.coded:
            tm = self.timing
            args = (
                depth_raw.data_ptr(), seg_raw.data_ptr(), c2w.data_ptr(), self.inv_intri_host.data_ptr(),
                poses.data_ptr(), poses.stride(0), self.range_gt.data_ptr(), self.voxel_size_gt.data_ptr(),
                self.gt_bits.data_ptr(), _lib.ptr(reset_mask), n, self.h, self.w, g, self.depth_sense_dist,
                self.prob_code.data_ptr(), self._tri_lut.data_ptr(), self.scanned_bits.data_ptr(), _lib.ptr(tri_out),
                int(tri_row_stride or 0), _lib.ptr(tri_i8_out), 0 if tri_i8_out is None else int(tri_i8_out.stride(0)),
                self.coverage_count.data_ptr(), self.code_overflow.data_ptr(), self.workspace.data_ptr(),
                self.workspace.numel(), 1 if (self.self_clean and not self._ws_dirty) else 0, _lib.stream_ptr(self.device))
            if tm is not None:
                tm.start()
            err = self.lib.gnbv_update_occ_grid_coded(*args)
            if tm is not None:
                tm.stop()
            _lib.check(err, "gnbv_update_occ_grid_coded")
            if not self.self_clean:
                self._ws_dirty = True  # (a call without the flag leaves the masks in place: never claim "clean" again)
            return tri_out
        if self.packed:
            _lib.check(self.lib.gnbv_update_occ_grid_packed(
                depth_raw.data_ptr(), seg_raw.data_ptr(), c2w.data_ptr(), self.inv_intri_host.data_ptr(),
                poses.data_ptr(), poses.stride(0), self.range_gt.data_ptr(), self.voxel_size_gt.data_ptr(),
                self.gt_bits.data_ptr(), _lib.ptr(reset_mask), n, self.h, self.w, g, self.depth_sense_dist,
                self._prob_f32.data_ptr(), self.scanned_bits.data_ptr(), tri_out.data_ptr(), int(tri_row_stride),
                self.coverage_count.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
                _lib.stream_ptr(self.device)), "gnbv_update_occ_grid_packed")
            return tri_out
        _lib.check(self.lib.gnbv_update_occ_grid(
            depth_raw.data_ptr(), seg_raw.data_ptr(), c2w.data_ptr(), self.inv_intri_host.data_ptr(),
            poses.data_ptr(), poses.stride(0), self.range_gt.data_ptr(), self.voxel_size_gt.data_ptr(),
            self.grid_gt.data_ptr(), _lib.ptr(reset_mask), n, self.h, self.w, g, self.depth_sense_dist,
            self._prob_f32.data_ptr(), self._scanned_f32.data_ptr(), tri_out.data_ptr(), int(tri_row_stride),
            self.coverage_count.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
            _lib.stream_ptr(self.device)), "gnbv_update_occ_grid")
        return tri_out

    @property
    def scanned_gt_grid(self) -> torch.Tensor:
        """The reference's fp32 [N,G,G,G] tensor (env_train_gennbv.py:182-183); in packed mode it is
        expanded from the bitmask on demand."""
        if not self.packed:
            return self._scanned_f32
        n, g = self.num_envs, self.grid_size
        out = torch.empty(n, g, g, g, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.gnbv_unpack_grid_bits(self.scanned_bits.data_ptr(), n, g, out.data_ptr(),
                                                  _lib.stream_ptr(self.device)), "gnbv_unpack_grid_bits")
        return out

    def masks(self):
        """(hit, path) bool [N,G,G,G] of the last update (parity / debugging).  The self-cleaning coded update leaves no masks
        behind: set `self_clean = False` before the update whose masks are wanted."""
        if self.coded and self.self_clean:
            raise _lib.GennbvHipError("masks(): the coded update cleared its masks (self_clean); set self_clean = False first")
        n, g = self.num_envs, self.grid_size
        hit = torch.empty(n, g, g, g, dtype=torch.uint8, device=self.device)
        path = torch.empty_like(hit)
        _lib.check(self.lib.gnbv_unpack_masks(self.workspace.data_ptr(), n, g, hit.data_ptr(), path.data_ptr(),
                                              _lib.stream_ptr(self.device)), "gnbv_unpack_masks")
        return hit.bool(), path.bool()
