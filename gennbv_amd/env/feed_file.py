"""On-disk container for a recorded simulator feed (the reference has none: it renders with
Isaac Gym every step).  One file = one scene block + F frames of what crosses the observation boundary
(gennbv/env/env_train_gennbv.py:346-354, :61-96):

    per frame : depth_raw [N,H,W] f16 or f32 (Isaac convention: negative metres, -inf = nothing),
                seg_raw [N,H,W] u8 (the reference reads it as float; values are small integers),
                view [N,4,4] f32 (Isaac view matrix), rgba [N,H,W,4] u8 (optional)
    scene     : grid_gt occupancy bits [N, G^3/8] (binary GT as loaded at :61-96), range_gt [N,6] f32,
                voxel_size [N,3] f32, num_valid_voxel_gt [N] f32, env_origins [N,3] f32

Layout: 16-byte preamble (magic "GNBVFEED", u32 version, u32 header bytes) + JSON header padded to 4 KiB +
the scene arrays + F equal-sized frame records, every array 64-byte aligned.  The header lists
(name, dtype, shape, offset) of the scene arrays and of the fields inside one frame record, so a reader maps
the file once (np.memmap) and views frame f without parsing.  Frames are appended; `close()` patches the
frame count.  f16 depth halves the file; it changes the input (11-bit mantissa: 1.6 cm at 20 m), so parity
fixtures use f32, recordings meant for throughput use f16 -- the decode to f32 happens once, on the GPU,
when the pool is made resident (`ReplayFeed.from_file`).
"""
from __future__ import annotations

import json
import os
import struct
from typing import Optional

import numpy as np
import torch

MAGIC = b"GNBVFEED"
VERSION = 1
_HEADER_PAD = 4096
_ALIGN = 64


def _align(x: int) -> int:
    return (x + _ALIGN - 1) // _ALIGN * _ALIGN


def _np(t) -> np.ndarray:
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


class FeedWriter:
    def __init__(self, path: str, num_envs: int, height: int, width: int, grid_size: int, depth_dtype: str = "f16",
                 with_rgba: bool = True):
        assert depth_dtype in ("f16", "f32")
        self.path, self.n, self.h, self.w, self.g = path, num_envs, height, width, grid_size
        self.depth_np = np.float16 if depth_dtype == "f16" else np.float32
        self.with_rgba = with_rgba
        self.frames = 0
        self._scene_written = False
        n, h, w, g = num_envs, height, width, grid_size
        assert g ** 3 % 8 == 0
        off = 0
        self.scene_fields = []
        for name, dt, shape in (("grid_gt_bits", "u1", (n, g ** 3 // 8)), ("range_gt", "f4", (n, 6)), ("voxel_size", "f4", (n, 3)),
                                ("num_valid_voxel_gt", "f4", (n,)), ("env_origins", "f4", (n, 3)), ("inv_intrinsics", "f4", (3, 3))):
            self.scene_fields.append({"name": name, "dtype": dt, "shape": list(shape), "offset": off})
            off = _align(off + int(np.prod(shape)) * np.dtype(dt).itemsize)
        self.scene_bytes = off
        off = 0
        self.frame_fields = []
        ff = [("depth_raw", "f2" if depth_dtype == "f16" else "f4", (n, h, w)), ("seg_raw", "u1", (n, h, w)), ("view", "f4", (n, 4, 4))]
        if with_rgba:
            ff.append(("rgba", "u1", (n, h, w, 4)))
        for name, dt, shape in ff:
            self.frame_fields.append({"name": name, "dtype": dt, "shape": list(shape), "offset": off})
            off = _align(off + int(np.prod(shape)) * np.dtype(dt).itemsize)
        self.frame_bytes = off
        self.f = open(path, "wb")
        self._write_header()
        self.f.seek(16 + _HEADER_PAD + self.scene_bytes)

    def _write_header(self):
        hdr = {"num_envs": self.n, "height": self.h, "width": self.w, "grid_size": self.g, "num_frames": self.frames,
               "scene_offset": 16 + _HEADER_PAD, "scene_bytes": self.scene_bytes, "scene_fields": self.scene_fields,
               "frames_offset": 16 + _HEADER_PAD + self.scene_bytes, "frame_bytes": self.frame_bytes, "frame_fields": self.frame_fields}
        blob = json.dumps(hdr).encode()
        assert len(blob) <= _HEADER_PAD
        pos = self.f.tell()
        self.f.seek(0)
        self.f.write(MAGIC + struct.pack("<II", VERSION, len(blob)) + blob + b" " * (_HEADER_PAD - len(blob)))
        self.f.seek(pos)

    def write_scene(self, grid_gt, range_gt, voxel_size, num_valid_voxel_gt, env_origins, inv_intrinsics) -> None:
        g = _np(grid_gt).reshape(self.n, -1)
        assert np.all((g == 0) | (g == 1)), "the container stores the reference's binary GT grid"
        vals = {"grid_gt_bits": np.packbits(g.astype(np.uint8), axis=1, bitorder="little"), "range_gt": _np(range_gt),
                "voxel_size": _np(voxel_size), "num_valid_voxel_gt": _np(num_valid_voxel_gt), "env_origins": _np(env_origins),
                "inv_intrinsics": _np(inv_intrinsics)}
        pos = self.f.tell()
        for fd in self.scene_fields:
            a = np.ascontiguousarray(vals[fd["name"]], dtype=np.dtype(fd["dtype"])).reshape(fd["shape"])
            self.f.seek(16 + _HEADER_PAD + fd["offset"])
            self.f.write(a.tobytes())
        self.f.seek(pos)
        self._scene_written = True

    def append(self, depth_raw, seg_raw, view, rgba=None) -> None:
        vals = {"depth_raw": _np(depth_raw).astype(self.depth_np), "seg_raw": _np(seg_raw), "view": _np(view)}
        seg = vals["seg_raw"]
        assert np.all((seg >= 0) & (seg <= 255) & (seg == np.floor(seg))), "seg_raw must hold integer class ids 0..255"
        if self.with_rgba:
            assert rgba is not None
            vals["rgba"] = _np(rgba)
        base = 16 + _HEADER_PAD + self.scene_bytes + self.frames * self.frame_bytes
        for fd in self.frame_fields:
            a = np.ascontiguousarray(vals[fd["name"]], dtype=np.dtype(fd["dtype"])).reshape(fd["shape"])
            self.f.seek(base + fd["offset"])
            self.f.write(a.tobytes())
        self.frames += 1
        self.f.truncate(base + self.frame_bytes)  # size the record (zero alignment padding at its end)

    def close(self) -> None:
        assert self._scene_written, "write_scene() was never called"
        self._write_header()
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class FeedFile:
    """Read-only memory map of a feed container."""

    def __init__(self, path: str):
        with open(path, "rb") as f:
            pre = f.read(16)
            if len(pre) < 16 or pre[:8] != MAGIC:
                raise ValueError(f"{path}: not a GNBVFEED container")
            version, hlen = struct.unpack("<II", pre[8:])
            if version != VERSION:
                raise ValueError(f"{path}: container version {version}, this reader understands {VERSION}")
            self.header = json.loads(f.read(hlen).decode())
        h = self.header
        self.num_envs, self.height, self.width = h["num_envs"], h["height"], h["width"]
        self.grid_size, self.num_frames = h["grid_size"], h["num_frames"]
        need = h["frames_offset"] + self.num_frames * h["frame_bytes"]
        if os.path.getsize(path) < need:
            raise ValueError(f"{path}: truncated ({os.path.getsize(path)} bytes, header promises {need})")
        self.mm = np.memmap(path, dtype=np.uint8, mode="r")

    def _view(self, base: int, fd) -> np.ndarray:
        dt = np.dtype(fd["dtype"])
        n = int(np.prod(fd["shape"])) * dt.itemsize
        return self.mm[base + fd["offset"]: base + fd["offset"] + n].view(dt).reshape(fd["shape"])

    def scene(self, name: str) -> np.ndarray:
        fd = next(f for f in self.header["scene_fields"] if f["name"] == name)
        return self._view(self.header["scene_offset"], fd)

    def frame(self, f: int, name: str) -> np.ndarray:
        if not 0 <= f < self.num_frames:
            raise IndexError(f)
        fd = next(x for x in self.header["frame_fields"] if x["name"] == name)
        return self._view(self.header["frames_offset"] + f * self.header["frame_bytes"], fd)

    @property
    def has_rgba(self) -> bool:
        return any(x["name"] == "rgba" for x in self.header["frame_fields"])

    def grid_gt(self) -> np.ndarray:
        """[N,G,G,G] f32 binary occupancy (what `_init_load_all` hands to the env, env_train_gennbv.py:61-96)."""
        g = self.grid_size
        bits = np.unpackbits(self.scene("grid_gt_bits"), axis=1, bitorder="little")[:, :g ** 3]
        return bits.reshape(self.num_envs, g, g, g).astype(np.float32)


def record(path: str, scene, frames, inv_intrinsics, depth_dtype: str = "f16", with_rgba: bool = True) -> None:
    """Write a synthetic.Scene + a list of synthetic.Frame (depth_raw, seg_raw, rgba, view) as one container."""
    n, h, w = frames[0].depth_raw.shape
    g = scene.grid_gt.shape[1]
    with FeedWriter(path, n, h, w, g, depth_dtype, with_rgba) as wr:
        wr.write_scene(scene.grid_gt, scene.range_gt, scene.voxel_size, scene.num_valid_voxel_gt, scene.env_origins, inv_intrinsics)
        for fr in frames:
            wr.append(fr.depth_raw, fr.seg_raw, fr.view, fr.rgba if with_rgba else None)


def load_scene(ff: FeedFile, device="cpu"):
    """synthetic.Scene equivalent of the container's scene block (boxes are not stored: empty)."""
    from . import synthetic as S
    t = lambda a: torch.from_numpy(np.array(a)).to(device)  # noqa: E731  (copy: the map is read-only)
    n = ff.num_envs
    return S.Scene(boxes_min=torch.zeros(n, 0, 3), boxes_max=torch.zeros(n, 0, 3), grid_gt=t(ff.grid_gt()), range_gt=t(ff.scene("range_gt")),
                   voxel_size=t(ff.scene("voxel_size")), num_valid_voxel_gt=t(ff.scene("num_valid_voxel_gt")),
                   env_origins=t(ff.scene("env_origins")))


def load_feed(ff: FeedFile, device, first: int = 0, count: Optional[int] = None):
    """Frames [first, first+count) as a HBM-resident ReplayFeed: the file's bytes go up as stored (f16 / u8),
    the widening to the f32 tensors the kernels read happens on the device."""
    from .replay_feed import ReplayFeed
    count = ff.num_frames - first if count is None else count
    up = lambda name, f: torch.from_numpy(np.array(ff.frame(f, name))).to(device)  # noqa: E731  (copy: the map is read-only)
    depth = torch.stack([up("depth_raw", f).float() for f in range(first, first + count)])
    seg = torch.stack([up("seg_raw", f).float() for f in range(first, first + count)])
    view = torch.stack([up("view", f) for f in range(first, first + count)])
    rgba = torch.stack([up("rgba", f) for f in range(first, first + count)]) if ff.has_rgba else None
    origins = torch.from_numpy(np.array(ff.scene("env_origins"))).to(device)
    return ReplayFeed.from_views(depth, seg, rgba, view, origins)
