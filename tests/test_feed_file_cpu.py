"""Replay-feed container (gennbv_amd/env/feed_file.py): what goes in comes out -- f32 depth bit
for bit (including -inf), f16 depth = the IEEE half rounding of the input, scene block and GT bits exact; the
reader rejects foreign / truncated files."""
import numpy as np
import pytest
import torch

from gennbv_amd.env import feed_file as FF
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig


def _recording(n=3, g=16, frames=4):
    cfg = TaskConfig(camera_width=32, camera_height=24, grid_size=g)
    scene = S.make_scenes(n, g, seed=5)
    fr = S.make_frames(scene, cfg, frames, seed=2)
    fr[0].depth_raw[0, 0, :4] = float("-inf")  # Isaac's "nothing rendered" value must survive both encodings
    return cfg, scene, fr


@pytest.mark.parametrize("depth_dtype", ["f32", "f16"])
def test_container_round_trip(tmp_path, depth_dtype):
    cfg, scene, fr = _recording()
    kinv = S.inverse_intrinsics(cfg.camera_height, cfg.camera_width, cfg.horizontal_fov)
    path = str(tmp_path / "feed.gnbv")
    FF.record(path, scene, fr, kinv, depth_dtype=depth_dtype)
    ff = FF.FeedFile(path)
    assert (ff.num_envs, ff.height, ff.width, ff.grid_size, ff.num_frames) == (3, 24, 32, 16, 4) and ff.has_rgba
    for f, frame in enumerate(fr):
        d = frame.depth_raw.numpy()
        assert f > 0 or np.isneginf(d).any()
        got = ff.frame(f, "depth_raw")
        if depth_dtype == "f32":
            assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), d.view(np.uint32))
        else:
            assert got.dtype == np.float16 and np.array_equal(got.view(np.uint16), d.astype(np.float16).view(np.uint16))
        assert np.array_equal(ff.frame(f, "seg_raw"), frame.seg_raw.numpy().astype(np.uint8))
        assert np.array_equal(ff.frame(f, "view").view(np.uint32), frame.view.numpy().view(np.uint32))
        assert np.array_equal(ff.frame(f, "rgba"), frame.rgba.numpy())
    assert np.array_equal(ff.grid_gt(), scene.grid_gt.numpy())
    for name in ("range_gt", "voxel_size", "num_valid_voxel_gt", "env_origins"):
        assert np.array_equal(ff.scene(name), getattr(scene, name).numpy().astype(np.float32)), name
    assert np.array_equal(ff.scene("inv_intrinsics"), kinv.numpy())
    sc = FF.load_scene(ff)
    assert torch.equal(sc.grid_gt, scene.grid_gt) and torch.equal(sc.range_gt, scene.range_gt.float())
    feed = FF.load_feed(ff, "cpu")
    ref = torch.stack([x.depth_raw for x in fr])
    assert torch.equal(feed.depth_raw, ref if depth_dtype == "f32" else ref.half().float())
    assert feed.c2w.shape == (4, 3, 4, 4) and feed.rgba.dtype == torch.uint8


def test_reader_rejects_foreign_and_truncated_files(tmp_path):
    bad = tmp_path / "x.bin"
    bad.write_bytes(b"NOTAFEED" + b"\0" * 64)
    with pytest.raises(ValueError):
        FF.FeedFile(str(bad))
    cfg, scene, fr = _recording(frames=2)
    path = str(tmp_path / "feed.gnbv")
    FF.record(path, scene, fr, S.inverse_intrinsics(24, 32, 90.0), depth_dtype="f16", with_rgba=False)
    assert not FF.FeedFile(path).has_rgba
    data = open(path, "rb").read()
    open(path, "wb").write(data[:-1000])
    with pytest.raises(ValueError):
        FF.FeedFile(path)
