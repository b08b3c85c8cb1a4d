"""ctypes binding of libgennbv_hip.so (C-ABI declared in include/gennbv_hip.h).

There is NO fallback: if the shared object is missing or a call fails, this
module raises.  The CPU oracle under oracle/ is test infrastructure and is
never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GENNBV_HIP_LIB", os.path.join(_HERE, "libgennbv_hip.so"))  # override: A/B builds

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float
_d = C.c_double
_sz = C.c_size_t

# name -> (restype, argtypes): must list every symbol include/gennbv_hip.h declares
SIGNATURES = {
    "gnbv_abi_version": (_i, []),
    "gnbv_build_arch": (C.c_char_p, []),
    "gnbv_post_process_depth": (_i, [_p, _p, _i64, _f, _p, _p, _p]),
    "gnbv_rgb_to_gray": (_i, [_p, _i, _i, _i, _i, _i, _p, _i64, _p]),
    "gnbv_back_projection": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "gnbv_points_to_idx": (_i, [_p, _p, _p, _p, _i, _i64, _i, _p, _p]),
    "gnbv_pose_to_idx": (_i, [_p, _p, _p, _i, _p, _p]),
    "gnbv_bresenham3d": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "gnbv_grid_tri_cls": (_i, [_p, _i64, _f, _f, _p, _p]),
    "gnbv_voxel_workspace_bytes": (_sz, [_i, _i]),
    "gnbv_voxel_workspace_bytes_hw": (_sz, [_i, _i, _i, _i]),
    "gnbv_update_occ_grid": (_i, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _i64, _p,
                                  _p, _sz, _p]),
    "gnbv_unpack_masks": (_i, [_p, _i, _i, _p, _p, _p]),
    "gnbv_grid_bit_words": (_i, [_i]),
    "gnbv_pack_grid_bits": (_i, [_p, _i, _i, _p, _p, _p]),
    "gnbv_unpack_grid_bits": (_i, [_p, _i, _i, _p, _p]),
    "gnbv_update_occ_grid_packed": (_i, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _i64, _p,
                                         _p, _sz, _p]),
    "gnbv_prob_code_tables": (None, [_p, _p]),
    "gnbv_decode_prob_grid": (_i, [_p, _i64, _p, _p, _p]),
    "gnbv_update_occ_grid_coded": (_i, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _i64, _p,
                                        _i64, _p, _p, _p, _sz, _i, _p]),
    "gnbv_env_pre_step": (_i, [_p, _p, _p, _i, _p, _p, _p]),
    "gnbv_env_obs_state": (_i, [_p, _p, _p, _p, _i, _i, _p, _i64, _p]),
    "gnbv_env_obs_rgb": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i64, _p]),
    "gnbv_env_observe": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _i64, _p, _p, _i, _i, _i, _i, _p, _p]),
    "gnbv_env_post_step": (_i, [_p, _p]),
    "gnbv_rollout_add": (_i, [_i, _i, _p, _p, _p, _p, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnbv_input_autocorr_row_ints": (_i, []),
    "gnbv_input_autocorr": (_i, [_p, _i64, _i, _i, _p, _i64, _p]),
    "gnbv_encoder_workspace_bytes": (_sz, [_i, _i]),
    "gnbv_encoder_y1_elems": (_sz, [_i, _i]),
    "gnbv_encoder_eval_prepare": (_i, [_i, _i, _p, _p, _p, _sz, _p]),
    "gnbv_encoder_grid_forward": (_i, [_p, _p, _i64, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnbv_encoder_grid_backward": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnbv_linear_workspace_bytes": (_sz, [_i, _i, _i]),
    "gnbv_linear_forward": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "gnbv_linear_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "gnbv_linear_bwd_prep": (_i, [_p, _p, _i, _i, _p, _p, _sz, _p]),
    "gnbv_linear_bwd_dx": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "gnbv_linear_bwd_dw": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "gnbv_linear_bwd_dw_sq_parts": (_i, [_i]),
    "gnbv_linear_bwd_dw_sq": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "gnbv_linear_fold_ok": (_i, [_i, _i, _i, _i]),
    "gnbv_linear_forward_fold": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "gnbv_linear_bwd_dw_fold": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "gnbv_pose_encode": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "gnbv_policy_head_forward": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    "gnbv_policy_head_backward": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnbv_gather_minibatch": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnbv_ppo_loss": (_i, [_p, _p]),
    "gnbv_ppo_loss_finish": (_i, [_p, _p]),
    "gnbv_multicategorical_sample": (_i, [_p, _i, _i, _i, _p, _p, _i, _p, _p, _p]),
    "gnbv_ppo_loss_rsl": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _f, _f, _i, _p, _p, _p, _p, _p]),
    "gnbv_adam_workspace_bytes": (_sz, []),
    "gnbv_clip_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _p, _p, _f, _p, _f, _p, _p, _sz, _p]),
    "gnbv_clip_adam_step_rotate": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _p, _p, _f, _p, _f, _p, _p, _sz, _p, _i, _i, _p, _p, _p]),
    "gnbv_clip_adam_step_ex": (_i, [_p, _p]),
    "gnbv_sq_partials_count": (_i, []),
    "gnbv_sq_partials": (_i, [_p, _i64, _p, _p]),
    "gnbv_adam_shard_step": (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _p, _p, _p]),
    "gnbv_chamfer_workspace_bytes": (_sz, [_i, _i]),
    "gnbv_chamfer_distance": (_i, [_p, _i, _p, _i, _p, _p, _sz, _p]),
    "gnbv_gae_sb3": (_i, [_p, _p, _p, _p, _p, _i, _i, _d, _d, _p, _p, _p]),
    "gnbv_gae_rsl": (_i, [_p, _p, _p, _p, _i, _i, _d, _d, _p, _p, _p]),
}


class GnbvLattice(C.Structure):
    """include/gennbv_hip.h: GnbvLattice"""
    _fields_ = [("clip_low", C.c_int64 * 6), ("clip_up", C.c_int64 * 6), ("init_action", C.c_int64 * 6),
                ("action_unit", C.c_float * 6), ("pose_low", C.c_float * 6), ("init_pose", C.c_float * 6)]


class GnbvEnvPost(C.Structure):
    """include/gennbv_hip.h: GnbvEnvPost"""
    _fields_ = [("n", _i), ("only_positive", _i), ("max_episode_length", _i64),
                ("scale_cov", _f), ("scale_short", _f), ("scale_term", _f), ("coverage_threshold", _f),
                ("coverage_count", _p), ("num_valid_voxel_gt", _p), ("prev_ratio", _p), ("episode_length_buf", _p),
                ("rewards", _p), ("dones", _p), ("reset_mask", _p), ("step_time_out", _p), ("extras_time_outs", _p),
                ("coverage_ratio", _p), ("episode_sums", _p), ("cur_reward_sum", _p), ("cur_episode_length", _p),
                ("ring_reward", _p), ("ring_length", _p), ("ring_state", _p), ("ring_len", _i), ("episode_info", _p), ("episode_state", _p),
                ("max_episode_length_s", _f)]


class GnbvEncoderParams(C.Structure):
    """include/gennbv_hip.h: GnbvEncoderParams"""
    _fields_ = [("w1", _p), ("b1", _p), ("bn1_w", _p), ("bn1_b", _p), ("bn1_rm", _p), ("bn1_rv", _p), ("bn1_nbt", _p),
                ("w2", _p), ("b2", _p), ("bn2_w", _p), ("bn2_b", _p), ("bn2_rm", _p), ("bn2_rv", _p), ("bn2_nbt", _p),
                ("eps", _f), ("momentum", _f), ("grid_i8", _p), ("grid_i8_row_stride", _i64),
                ("autocorr", _p), ("autocorr_row_stride", _i64),
                ("world", _i), ("sync_sum", _p), ("sync_ctx", _p), ("sync_buf", _p), ("autocorr_global", _p),
                ("autocorr_total", _p), ("force_fp32", _i), ("range_flag", _p), ("eval_prepared", _i)]


class GnbvAdamStep(C.Structure):
    """include/gennbv_hip.h: GnbvAdamStep"""
    _fields_ = [("params", _p), ("grads", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("n", _i64),
                ("max_grad_norm", _f), ("lr", _f), ("beta1", _f), ("beta2", _f), ("eps", _f),
                ("step", _p), ("stop_flag", _p), ("grad_scale", _f), ("kl_slot", _p), ("target_kl", _f),
                ("norm_out", _p), ("workspace", _p), ("workspace_bytes", _sz),
                ("table", _p), ("table_rows", _i), ("row_len", _i), ("out", _p), ("counter", _p),
                ("sq_lo", _i64), ("sq_hi", _i64), ("sq_partial", _p), ("sq_parts", _i), ("loss_finish", _p), ("upd_skip_lo", _i64), ("upd_skip_hi", _i64)]


class GnbvEncoderGrads(C.Structure):
    """include/gennbv_hip.h: GnbvEncoderGrads"""
    _fields_ = [(k, _p) for k in ("w1", "b1", "bn1_w", "bn1_b", "w2", "b2", "bn2_w", "bn2_b")]


class GnbvPpoLoss(C.Structure):
    """include/gennbv_hip.h: GnbvPpoLoss"""
    _fields_ = [("batch", _i), ("n_logits", _i), ("n_heads", _i), ("head_dims", _i * 8), ("normalize_advantage", _i),
                ("clip_range", _f), ("clip_range_vf", _f), ("ent_coef", _f), ("vf_coef", _f), ("policy_scale", _f),
                ("target_kl", _f),
                ("logits", _p), ("values", _p), ("actions", _p), ("old_values", _p), ("old_log_prob", _p),
                ("advantages", _p), ("returns", _p), ("d_logits", _p), ("d_values", _p), ("head_entropy", _p),
                ("head_lse", _p), ("stats", _p), ("stats_row", _p), ("stop_flag", _p), ("scratch", _p), ("kl_out", _p),
                ("rows", _p), ("adv_norm", _p), ("defer_stats", _i)]


class GennbvHipError(RuntimeError):
    pass


_lib = None
_loaded = {}


def _open(path: str, strict: bool = True):
    if path in _loaded:
        return _loaded[path]
    if not os.path.exists(path):
        raise GennbvHipError(
            f"{path} is missing: build it with `python -m gennbv_amd.csrc.build` "
            "(or __graft_entry__.build()).  gennbv_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        except AttributeError:
            if strict:
                raise
            continue  # (activate(): an OLDER build of the same ABI under A/B may lack entry points added since)
        fn.restype = res
        fn.argtypes = args
    if lib.gnbv_abi_version() != 5:
        raise GennbvHipError("libgennbv_hip.so ABI version mismatch")
    _loaded[path] = lib
    return lib


def load():
    """dlopen the HIP library and bind every C-ABI symbol.  Raises if it is absent."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def activate(path=None):
    """A/B tooling only (tools/ab_interleaved.py): make `path` (default: the in-tree library) the library load() returns from now on.
    Two builds can live in one process -- kernels are bound when a call / a hipGraph capture is made, so objects built and graphs
    captured while a library was active keep running its kernels."""
    global _lib
    _lib = _open(os.path.abspath(path), strict=False) if path else _open(LIB_PATH)
    return _lib


def check(err: int, what: str):
    if err != 0:
        raise GennbvHipError(f"{what} failed with hipError_t={err}")


def ptr(t):
    """Device (or host) pointer of a tensor; None -> NULL."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev_idx = {}


def stream_ptr(device=None):
    """The caller's current HIP stream as void* (the kernels enqueue on it).  Through torch's raw-stream accessor: the
    `torch.cuda.current_stream(device).cuda_stream` it replaces builds a Stream object per call -- 15 calls and ~75 us of HOST time per
    rollout step, which is launch-bound (tools/profile_rollout_host.py)."""
    if _raw_stream is None:
        return torch.cuda.current_stream(device).cuda_stream
    idx = _dev_idx.get(device)
    if idx is None:
        d = torch.device(device) if device is not None else None
        idx = d.index if (d is not None and d.index is not None) else None
        if idx is not None and not isinstance(device, torch.Tensor):
            try:
                _dev_idx[device] = idx
            except TypeError:
                pass
    return _raw_stream(torch.cuda.current_device() if idx is None else idx)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GennbvHipError("gennbv_amd operators run on the GPU only (tensor is on %s); "
                                 "there is no CPU fallback" % t.device)


def require_contig(*tensors):
    for t in tensors:
        if t is not None and not t.is_contiguous():
            raise GennbvHipError("tensor must be contiguous")
