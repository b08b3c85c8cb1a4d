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

    def __init__(self, world: int, group=None, always_sync: bool = False, side_group=None):
        self.world, self.group = int(world), group
        # exercise the collective path even with one rank (single-GPU test of the data-parallel code)
        self.active = self.world > 1 or always_sync
        self.sync_buf = None   # fp64 [96] on the device: the three 32-double sums of GnbvEncoderParams.sync_sum
        self._cb = None
        # Round 6: everything that is exchanged EAGERLY -- attach()'s broadcasts, the per-train() statistics tables, the
        # range-guard flag -- goes over a gloo group of the same ranks on host copies, so that an RCCL communicator carries nothing but the
        # captured step (+ its one-off warm-up): a process then no longer mixes eager and captured collectives call after call, which is
        # what the process-group watchdog abort of round 5 needed.  None: the main group is used (gloo tests, world 1).
        self.side_group = side_group

    def _side(self, t: torch.Tensor, fn) -> torch.Tensor:
        """Run the eager collective `fn(tensor, group)` on the side group (host copy) when there is one, else on the main group in place."""
        if self.side_group is None:
            fn(t, self.group)
            return t
        h = t.detach().to("cpu")
        fn(h, self.side_group)
        t.copy_(h.to(t.device))
        return t

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> None:
        self._side(t, lambda x, g: dist.broadcast(x, src=src, group=g))

    def all_reduce_eager_(self, t: torch.Tensor, op=None) -> torch.Tensor:
        op = dist.ReduceOp.SUM if op is None else op
        return self._side(t, lambda x, g: dist.all_reduce(x, op=op, group=g))

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
        self.all_reduce_eager_(s1)
        mean = s1 / bg
        s2 = ((a - mean[:, None]) ** 2).sum(1)
        self.all_reduce_eager_(s2)
        std = torch.sqrt(s2 / max(bg - 1.0, 1.0))
        return torch.stack((mean, 1.0 / (std + 1e-8)), dim=1).float().contiguous()

    def global_autocorr(self, ac_rows: torch.Tensor) -> torch.Tensor:
        """ac_rows [n_mb, B, 768] int32 (this rank's autocorrelation rows in minibatch order) -> [n_mb, 768] int32 totals of
        the global minibatches."""
        tot = ac_rows.sum(1, dtype=torch.int64).to(torch.int32).contiguous()
        return self.all_reduce_eager_(tot)

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
            self.all_reduce(flat_with_slot)

    # ---- the collectives of the fused step (PPO_Grid_Obs._dp_step_body): thin wrappers, so that the step can run WITHOUT a process group
    # (`null`: one rank, every exchange the identity) -- what the replay-order stress test drives: the stream / allocator ordering of the
    # data-parallel step with the host enqueued far ahead of the device, no RCCL needed
    null = False

    class _Done:
        """Work handle of a null / synchronous exchange.  Like a process group's work, `wait()` makes the CALLER's current stream wait for the
        stream the exchange was issued on (an event recorded there) -- inside a capture that is what joins the second stream's branch."""

        def __init__(self, event=None):
            self.event = event

        def wait(self):
            if self.event is not None:
                torch.cuda.current_stream().wait_event(self.event)
            return True

    def _done_here(self, t: torch.Tensor):
        if not t.is_cuda:
            return self._Done()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        return self._Done(ev)

    def all_reduce(self, t: torch.Tensor, async_op: bool = False):
        if self.null:
            return self._done_here(t) if async_op else self._Done()
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return w if async_op else self._Done()

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        if self.null:
            out.copy_(inp)  # (one rank owns the whole slice)
            return self._done_here(out) if async_op else self._Done()
        w = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return w if async_op else self._Done()

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        if self.null:
            out.copy_(inp)
            return
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def rank(self) -> int:
        return 0 if self.null else dist.get_rank(self.group)

    def capturable(self):
        """(can the collectives be recorded into a hipGraph?, backend name)"""
        if self.null:
            return True, "null"
        backend = str(dist.get_backend(self.group)).lower()
        return "nccl" in backend, backend


def attach_null(algo) -> GradSync:
    """The data-parallel code path of `algo` with ONE rank and NO process group: every exchange of the fused step is the identity.  For
    tests of the step's stream / allocator ordering (tests/test_ppo_gpu.py: replay-order stress) -- not a way to train."""
    sync = GradSync(1, None, always_sync=True)
    sync.null = True
    algo._sync = sync
    return sync


def exchange_probe(algo, iters: int = 10) -> dict:
    """Raw duration of the step's exchanges, back to back with nothing overlapping them (events on the current stream; call it behind the
    timed region, every rank alike): the reduce-scatter and the all-gather of fc_grid.weight's slice, the all-reduce of the rest and of
    the conv gradients + KL slot.  With the bytes and the world size the figures answer "ring or direct" for the first real multi-GPU
    run by themselves (a ring over 7 xGMI links is per-link bound at ~0.66 ms for 58 MB, a direct exchange at ~0.1 ms)."""
    st, sync = algo._hip, algo._sync
    if not st or sync is None or not sync.active or sync.null:
        return {}
    opt, n_conv = st["opt"], st["n_conv"]
    sh = getattr(opt, "shard", None)
    ops = {}
    if sh is not None:
        lo, hi = sh["lo"], sh["hi"]
        scratch = torch.empty_like(opt.grads[lo:hi])
        p_shard = torch.empty_like(sh["grad"])
        ops["reduce_scatter_fc_grid_weight"] = (lambda: sync.reduce_scatter(sh["grad"], scratch), scratch.numel() * 4)
        ops["all_gather_fc_grid_weight"] = (lambda: sync.all_gather(scratch, p_shard), scratch.numel() * 4)
        rest = torch.empty_like(opt.grads[hi:])
        ops["all_reduce_rest_of_late_gradients"] = (lambda: sync.all_reduce(rest), rest.numel() * 4)
    else:
        late = torch.empty_like(opt.grads_with_slot[opt.SLOT + n_conv:])
        ops["all_reduce_late_gradients"] = (lambda: sync.all_reduce(late), late.numel() * 4)
    conv = torch.empty_like(opt.grads_with_slot[:opt.SLOT + n_conv])
    ops["all_reduce_conv_gradients_and_kl_slot"] = (lambda: sync.all_reduce(conv), conv.numel() * 4)
    out = {"world": sync.world, "backend": sync.capturable()[1], "iters": iters,
           "env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")}}
    for name, (fn, nbytes) in ops.items():
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        out[name] = {"bytes": int(nbytes), "ms": e0.elapsed_time(e1) / iters}
    return out


def _is_nccl(group) -> bool:
    return dist.is_initialized() and "nccl" in str(dist.get_backend(group)).lower()


def attach(algo, world: int, group=None, always_sync: bool = False, side_group="auto") -> GradSync:
    """Make `algo` (PPO_Grid_Obs) a data-parallel replica: identical initial parameters on every
    rank (broadcast from rank 0) and gradient / KL synchronisation in train().

    side_group="auto": with an RCCL main group and more than one rank, a gloo group of the same ranks is created HERE (a collective call:
    every rank attaches) for the eager exchanges -- see GradSync.__init__; pass a group to use yours, None to keep everything on the main one."""
    if side_group == "auto":
        side_group = None
        if world > 1 and _is_nccl(group):
            ranks = None if group is None else dist.get_process_group_ranks(group)
            try:
                side_group = dist.new_group(ranks=ranks, backend="gloo")
            except Exception as ex:  # (no usable host interface for gloo: the same on every rank -- the eager exchanges stay on the main group)
                import warnings
                warnings.warn(f"[gennbv_amd] no gloo side group ({ex!r}): attach()'s broadcasts and the per-train() tables use the RCCL group")
                side_group = None
    sync = GradSync(world, group, always_sync, side_group)
    algo._sync = sync
    if sync.active and _is_nccl(group):
        import warnings
        for k in ("TORCH_NCCL_CUDA_EVENT_CACHE", "TORCH_NCCL_TRACE_BUFFER_SIZE"):
            if os.environ.get(k) != "0":
                warnings.warn(f"[gennbv_amd] {k} is not 0: this process group was not created through parallel.init_process_group / "
                              "capture_safe_env(); a process that captures RCCL collectives into hipGraphs should run with it (see its docstring)")
    if sync.active and world > 1:  # (one rank: the broadcast is the identity)
        for t in list(algo.policy.parameters()) + list(algo.policy.buffers()):
            sync.broadcast_(t.data, src=0)
    return sync
