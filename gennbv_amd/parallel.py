"""Multi-GPU: environments shard across ranks, the policy is replicated, gradients are averaged
with ONE all-reduce per optimizer step over RCCL/xGMI.

The reference has no multi-GPU path (no torch.distributed / NCCL call site anywhere), so this
module defines it:

  * one process per GPU, `torch.distributed` with backend "nccl" (= RCCL on ROCm); rank r owns
    envs [r*N, (r+1)*N) with their grids, pose history, feed and rollout-buffer columns --
    state encoding, buffer add and the GAE scan need no exchange (independent per env);
  * every rank draws the same minibatch permutation (same numpy seed) over its own columns,
    so the global minibatch k is the concatenation of the ranks' minibatches k;
  * the FLAT gradient buffer is all-reduced (sum) once per optimizer step, in two pieces: the late
    layers (fc_grid: 55 MB of the 58 MB at G=64) as soon as their gradients exist, overlapped with
    the conv-stack backward, then the small conv-stack piece; that second piece carries one extra
    slot with the rank's approx-KL, so the early-stop decision (KL of the global minibatch = mean
    of the ranks' KLs) rides in the collective and every rank stops at the same minibatch -- no
    divergence, no deadlock;
  * clip_grad_norm_ is evaluated on the averaged gradient (after the all-reduce);
  * the statistics that the single-GPU update takes over the MINIBATCH are taken over the GLOBAL minibatch (the union of
    the ranks' minibatches k), so that a data-parallel update equals the single-GPU update of the global batch:
      - advantage mean / unbiased std (ppo_grid_obs.py:214-216): a table [n_minibatches, 2] computed once per train() --
        the advantages and the permutation are fixed for all epochs -- with two small all-reduces (`global_adv_norm`);
      - BatchNorm-1 batch statistics: analytic from the input autocorrelation total of the global minibatch, also a
        per-train() table (`global_autocorr`: [n_minibatches, 768] int32, one all-reduce);
      - BatchNorm-2 batch sums (forward) and the BatchNorm-2 / BatchNorm-1 backward sums: 32 doubles each, summed over
        the ranks inside the encoder calls through `GnbvEncoderParams.sync_sum` (three small all-reduces per optimizer
        step, captured in the step's hipGraph like the gradient all-reduce).
    Parameter gradients stay local sums; the gradient all-reduce adds them (see csrc/encoder.hip, k_c1w_fused_finish).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 58 MB fp32 gradient of the G=64
policy costs ~2*(7/8)*58 MB / (7*153 GB/s) = 95 us on a direct reduce-scatter + all-gather
when every link is used, ~0.66 ms on a single ring; RCCL picks the algorithm.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def capture_safe_env() -> None:
    """Environment defaults for a process that mixes EAGER and hipGraph-CAPTURED RCCL collectives (every data-parallel run here: attach()
    broadcasts and the warm-up steps are eager, the step itself is captured).  Call before the process group is created.

    Round 5: two processes in a few hundred died in ProcessGroupNCCL's watchdog thread with "operation not permitted on an event last
    recorded in a capturing stream": the watchdog (and its flight recorder, which times every collective through the work's events) queries
    events of eager works while events from the same cache are being recorded by captured works.  Without the event cache a captured work
    never records an event an eager work's bookkeeping still knows, and without the trace buffer nothing but completion is queried.
    (`setdefault`: a caller's own setting wins.  Not reproducible on demand, so this is a precaution, not a proven fix; the capture itself
    also waits for the watchdog to drain -- PPO_Grid_Obs._capture_minibatch_graph.)"""
    os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
    os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "0")


def init_process_group(backend: Optional[str] = None) -> int:
    """RANK / WORLD_SIZE / MASTER_* come from the launcher (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts
        capture_safe_env()
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend)
    return world


def shard_range(total_envs: int, rank: int, world: int):
    """Contiguous env shard of `rank`."""
    assert total_envs % world == 0, "envs must divide evenly over the ranks"
    per = total_envs // world
    return rank * per, (rank + 1) * per


class GradSync:
    """Gradient (+ KL) averaging for both train paths, and the global-minibatch statistics of the fused path."""

    def __init__(self, world: int, group=None, always_sync: bool = False):
        self.world, self.group = int(world), group
        # exercise the collective path even with one rank (single-GPU test of the data-parallel code)
        self.active = self.world > 1 or always_sync
        self.sync_buf = None   # fp64 [96] on the device: the three 32-double sums of GnbvEncoderParams.sync_sum
        self._cb = None

    # ---- global-minibatch statistics (fused path) ---------------------------------------
    def encoder_sync(self, device):
        """(callback pointer, sync_buf) for GnbvEncoderParams: sums sync_buf[offset : offset + n] over the ranks, enqueued on
        the caller's current stream (the stream the encoder kernels run on)."""
        import ctypes as C
        if self.sync_buf is None:
            self.sync_buf = torch.zeros(96, dtype=torch.float64, device=device)
            buf, group = self.sync_buf, self.group

            def cb(_ctx, offset, n, _stream):
                try:
                    dist.all_reduce(buf[offset:offset + n], op=dist.ReduceOp.SUM, group=group)
                    return 0
                except Exception as ex:  # surfaces as a failed encoder call
                    print(f"[gennbv_amd] sync_sum failed: {ex!r}")
                    return 1
            self._cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)(cb)
        return C.cast(self._cb, C.c_void_p).value, self.sync_buf

    def global_adv_norm(self, adv_rows: torch.Tensor) -> torch.Tensor:
        """adv_rows [n_mb, B] = this rank's advantages in minibatch order -> [n_mb, 2] fp32 (mean, 1 / (std + 1e-8)) of the
        global minibatches (unbiased std, ppo_grid_obs.py:214-216).  Two-pass in fp64."""
        a = adv_rows.double()
        bg = float(a.shape[1] * self.world)
        s1 = a.sum(1)
        dist.all_reduce(s1, op=dist.ReduceOp.SUM, group=self.group)
        mean = s1 / bg
        s2 = ((a - mean[:, None]) ** 2).sum(1)
        dist.all_reduce(s2, op=dist.ReduceOp.SUM, group=self.group)
        std = torch.sqrt(s2 / max(bg - 1.0, 1.0))
        return torch.stack((mean, 1.0 / (std + 1e-8)), dim=1).float().contiguous()

    def global_autocorr(self, ac_rows: torch.Tensor) -> torch.Tensor:
        """ac_rows [n_mb, B, 768] int32 (this rank's autocorrelation rows in minibatch order) -> [n_mb, 768] int32 totals of
        the global minibatches."""
        tot = ac_rows.sum(1, dtype=torch.int64).to(torch.int32).contiguous()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        return tot

    # ---- torch-module path (per-parameter grads) --------------------------------------
    def average_grads(self, params: Iterable[torch.nn.Parameter]) -> None:
        if not self.active:
            return
        grads = [p.grad for p in params if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def mean_scalar(self, x: torch.Tensor) -> torch.Tensor:
        if not self.active:
            return x
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y / self.world

    # ---- fused path: flat gradient buffer with the KL slot appended ---------------------
    def all_reduce_flat(self, flat_with_slot: torch.Tensor) -> None:
        if self.active:
            dist.all_reduce(flat_with_slot, op=dist.ReduceOp.SUM, group=self.group)


def attach(algo, world: int, group=None, always_sync: bool = False) -> GradSync:
    """Make `algo` (PPO_Grid_Obs) a data-parallel replica: identical initial parameters on every
    rank (broadcast from rank 0) and gradient / KL synchronisation in train()."""
    sync = GradSync(world, group, always_sync)
    algo._sync = sync
    if sync.active:
        for t in list(algo.policy.parameters()) + list(algo.policy.buffers()):
            dist.broadcast(t.data, src=0, group=group)
    return sync
