"""torch.autograd bridge to the gfx950 encoder kernels (gennbv_amd/csrc/encoder.hip).

`grid_encoder(obs, rows, module, training)` evaluates
`naive_encoder_grid(obs[rows, s:s+G^3].reshape(B,1,G,G,G)).reshape(B,-1)` of the reference
(gennbv/network/hybrid_encoder.py:90-94) -- conv1, BN1, ReLU, conv2, BN2, ReLU -- and its
backward on hand-written kernels through the C-ABI.  `rows` (int64 [B] or None) selects the
minibatch rows straight out of the rollout buffer, so the gather is fused into conv1's load.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from .. import _lib


def _flat_rows(small: torch.Tensor, grid_i8: torch.Tensor, state_dim: int) -> torch.Tensor:
    """compact rows [state | state_rgb] + int8 grid -> the reference's flat fp32 rows [state | grid | state_rgb]."""
    return torch.cat((small[:, :state_dim], grid_i8.to(torch.float32), small[:, state_dim:]), dim=1)


class RowGather:
    """A lazily gathered observation batch: rows `rows` of the 2-D fp32 matrix `base`.

    `grid_i8` ([R, G^3] int8, same row numbering): compact copy of the tri-class grid slices; the conv1 kernels read
    it instead of the fp32 slice of `base`.  `compact_state_dim = s`: COMPACT observations -- `base` rows are
    [state (s) | state_rgb] only, the grid exists solely as `grid_i8` (a quarter of the bytes of a flat fp32 row)."""

    def __init__(self, base: torch.Tensor, rows: torch.Tensor, grid_i8: Optional[torch.Tensor] = None,
                 compact_state_dim: Optional[int] = None, autocorr: Optional[torch.Tensor] = None):
        assert base.dim() == 2 and base.stride(1) == 1 and rows.dtype == torch.int64
        # per-row input autocorrelation of the int8 grid rows (input_autocorr(): [R, 768] int32, same row numbering)
        assert autocorr is None or (grid_i8 is not None and autocorr.dtype == torch.int32 and autocorr.shape[0] == base.shape[0]
                                    and autocorr.stride(1) == 1)
        self.autocorr = autocorr
        self.base, self.rows = base, rows.contiguous()
        assert grid_i8 is None or (grid_i8.dtype == torch.int8 and grid_i8.dim() == 2 and grid_i8.stride(1) == 1
                                   and grid_i8.shape[0] == base.shape[0])
        assert compact_state_dim is None or grid_i8 is not None
        self.grid_i8, self.compact_state_dim = grid_i8, compact_state_dim
        self.shape = (rows.shape[0], base.shape[1] + (0 if compact_state_dim is None else grid_i8.shape[1]))
        self.device, self.is_cuda = base.device, base.is_cuda

    def float(self):
        return self

    def materialize(self) -> torch.Tensor:
        if self.compact_state_dim is not None:
            return _flat_rows(self.base[self.rows], self.grid_i8[self.rows], self.compact_state_dim)
        return self.base[self.rows]

    def columns(self, a: int, b: int) -> torch.Tensor:
        assert self.compact_state_dim is None or b <= self.compact_state_dim
        return self.base[:, a:b][self.rows]


class DenseObs:
    """A whole observation matrix [N, D_obs] together with the compact int8 copy of its grid slices [N, G^3]
    (rollout forward: every row is used, no gather).  `compact_state_dim`: see RowGather."""

    def __init__(self, base: torch.Tensor, grid_i8: Optional[torch.Tensor] = None, compact_state_dim: Optional[int] = None):
        assert base.dim() == 2 and base.stride(1) == 1
        assert grid_i8 is None or (grid_i8.dtype == torch.int8 and grid_i8.shape[0] == base.shape[0] and grid_i8.stride(1) == 1)
        assert compact_state_dim is None or grid_i8 is not None
        self.base, self.rows, self.grid_i8, self.compact_state_dim = base, None, grid_i8, compact_state_dim
        self.autocorr = None  # (rollout forward: no backward, no autocorrelation needed)
        self.shape = (base.shape[0], base.shape[1] + (0 if compact_state_dim is None else grid_i8.shape[1]))
        self.device, self.is_cuda = base.device, base.is_cuda

    def float(self):
        return self

    def materialize(self) -> torch.Tensor:
        if self.compact_state_dim is not None:
            return _flat_rows(self.base, self.grid_i8, self.compact_state_dim)
        return self.base

    def columns(self, a: int, b: int) -> torch.Tensor:
        assert self.compact_state_dim is None or b <= self.compact_state_dim
        return self.base[:, a:b]


def conv_out(g: int) -> int:
    return (g - 3) // 2 + 1


_ws_cache = {}


def _workspace(lib, batch, grid, device):
    key = (batch, grid, str(device))
    ws = _ws_cache.get(key)
    if ws is None:
        n = lib.gnbv_encoder_workspace_bytes(batch, grid)
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        assert ws.data_ptr() % 256 == 0
        _ws_cache[key] = ws
    return ws


def input_autocorr(grid_i8: torch.Tensor, grid: int, out: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    """Per-row autocorrelation of the conv1 input patches of int8 grid rows [n, G^3] -> [n, 768] int32
    (include/gennbv_hip.h: gnbv_input_autocorr).  `stream`: raw HIP stream to issue on (default: the current stream)."""
    lib = _lib.load()
    _lib.require_cuda(grid_i8)
    n = grid_i8.shape[0]
    assert grid_i8.dtype == torch.int8 and grid_i8.dim() == 2 and grid_i8.stride(1) == 1 and grid_i8.shape[1] >= grid ** 3
    if out is None:
        out = torch.empty(n, lib.gnbv_input_autocorr_row_ints(), dtype=torch.int32, device=grid_i8.device)
    assert out.dtype == torch.int32 and out.shape[0] == n and out.stride(1) == 1
    _lib.check(lib.gnbv_input_autocorr(grid_i8.data_ptr(), grid_i8.stride(0), n, grid, out.data_ptr(), out.stride(0),
                                       _lib.stream_ptr(grid_i8.device) if stream is None else stream), "gnbv_input_autocorr")
    return out


def autocorr_supported(grid: int) -> bool:
    return grid >= 16 and grid % 16 == 0 and 3 * grid * grid <= 64 * 1024


def _params_struct(seq, grid_i8: Optional[torch.Tensor] = None,
                   autocorr: Optional[torch.Tensor] = None, dp=None, guard=None) -> _lib.GnbvEncoderParams:
    conv1, bn1, conv2, bn2 = seq[0], seq[1], seq[3], seq[4]
    p = _lib.GnbvEncoderParams()
    p.w1, p.b1, p.bn1_w, p.bn1_b = conv1.weight.data_ptr(), conv1.bias.data_ptr(), bn1.weight.data_ptr(), bn1.bias.data_ptr()
    p.bn1_rm, p.bn1_rv, p.bn1_nbt = bn1.running_mean.data_ptr(), bn1.running_var.data_ptr(), bn1.num_batches_tracked.data_ptr()
    p.w2, p.b2, p.bn2_w, p.bn2_b = conv2.weight.data_ptr(), conv2.bias.data_ptr(), bn2.weight.data_ptr(), bn2.bias.data_ptr()
    p.bn2_rm, p.bn2_rv, p.bn2_nbt = bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(), bn2.num_batches_tracked.data_ptr()
    p.eps, p.momentum = float(bn1.eps), float(bn1.momentum)
    p.grid_i8 = None if grid_i8 is None else grid_i8.data_ptr()
    p.grid_i8_row_stride = 0 if grid_i8 is None else int(grid_i8.stride(0))
    p.autocorr = None if autocorr is None else autocorr.data_ptr()
    p.autocorr_row_stride = 0 if autocorr is None else int(autocorr.stride(0))
    # data-parallel replicas: BatchNorm over the global minibatch (GnbvEncoderParams.world; gennbv_amd/parallel.py)
    p.world = 0 if dp is None else int(dp["world"])
    p.sync_sum = None if dp is None else dp["cb"]
    p.sync_ctx = None
    p.sync_buf = None if dp is None else dp["sync_buf"].data_ptr()
    p.autocorr_global = None if dp is None else dp["autocorr_global"].data_ptr()
    # operand ranges of the split-f16 kernels (network/hybrid_encoder.py: check_operand_ranges): (force_fp32, range_flag tensor)
    p.force_fp32 = 0 if guard is None else int(bool(guard[0]))
    p.range_flag = None if guard is None or guard[1] is None else guard[1].data_ptr()
    # (guard[2], optional: int32 [768] device tensor with the minibatch's autocorrelation total -- GnbvEncoderParams.autocorr_total)
    p.autocorr_total = None if guard is None or len(guard) < 3 or guard[2] is None else guard[2].data_ptr()
    return p


class _NoCtx:
    """Stand-in for the autograd context when no graph is recorded (rollout, evaluation): `Function.apply` costs ~35 us of host time per
    call even under no_grad (functorch wrappers, context set-up) -- five calls per rollout step, a third of its host time, and the step
    is launch-bound (tools/profile_rollout_host.py).  The forward runs unchanged; what it hands to the context is dropped."""

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def _run(fn, *args):
    """fn.apply(*args) when autograd records, fn.forward on a throw-away context otherwise (same code, same kernels)."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoCtx(), *args)


class _GridEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, rows, grid_off, grid, training, skip_flag, seq, write_through, grid_i8, compact, autocorr, dp, guard, fold, w1, b1, g1, be1, w2, b2, g2, be2):
        lib = _lib.load()
        _lib.require_cuda(base, w1)
        for t in (w1, b1, g1, be1, w2, b2, g2, be2):
            assert t.is_contiguous() and t.dtype == torch.float32
        dev = base.device
        batch = int(rows.shape[0]) if rows is not None else int(base.shape[0])
        o1 = conv_out(grid)
        o2 = conv_out(o1)
        p2 = o2 ** 3
        y1 = torch.empty(lib.gnbv_encoder_y1_elems(batch, grid), dtype=torch.float32, device=dev)
        y2 = torch.empty(batch * 16 * p2, dtype=torch.float32, device=dev)
        bn_state = torch.empty(2 * 4 * 16 + 768, dtype=torch.float32, device=dev)  # + the minibatch's autocorrelation total (ints)
        # fold: BatchNorm-2 + ReLU are left to the consumer's operand load (linear_relu(..., fold=...)): no feature tensor; the
        # outputs are the raw conv output y2 viewed as [B, 16 P2] and bn_state (its floats 64..96 = BN2's scale | shift)
        feats = None if fold else torch.empty(batch, 16 * p2, dtype=torch.float32, device=dev)
        ws = _workspace(lib, batch, grid, dev)
        params = _params_struct(seq, grid_i8, autocorr, dp if training else None, guard)
        # compact observations: `base` has no grid slice, the kernels read the int8 rows only (obs pointer NULL)
        assert not compact or grid_i8 is not None
        obs_ptr = None if compact else base.data_ptr() + 4 * grid_off
        _lib.check(lib.gnbv_encoder_grid_forward(
            obs_ptr, _lib.ptr(rows), base.stride(0), batch, grid, C.byref(params), int(training), _lib.ptr(skip_flag),
            y1.data_ptr(), y2.data_ptr(), bn_state.data_ptr(), _lib.ptr(feats), ws.data_ptr(), ws.numel(),
            _lib.stream_ptr(dev)), "gnbv_encoder_grid_forward")
        ctx.save_for_backward(base, rows, y1, y2, bn_state, w1, w2)
        ctx.meta = (grid_off, grid, batch, seq)
        ctx.write_through = write_through
        ctx.grid_i8 = grid_i8
        ctx.autocorr = autocorr
        ctx.dp = dp if training else None
        ctx.guard = guard
        ctx.obs_ptr = obs_ptr
        if fold:
            # (the "gradient of this output" that comes back is d loss / d relu(bn2(y2)): the consumer's input gradient)
            ctx.mark_non_differentiable(bn_state)
            ctx.set_materialize_grads(False)  # (or autograd fills a zero "gradient" of bn_state: a launch on the critical path)
            return y2.view(batch, 16 * p2), bn_state
        return feats

    @staticmethod
    def backward(ctx, d_feats, _d_bn_state=None):
        lib = _lib.load()
        base, rows, y1, y2, bn_state, w1, w2 = ctx.saved_tensors
        grid_off, grid, batch, seq = ctx.meta
        dev = base.device
        o1 = conv_out(grid)
        o2 = conv_out(o1)
        d_feats = d_feats.contiguous().float()
        dy2 = torch.empty(batch * o2 ** 3 * 16, dtype=torch.float32, device=dev)
        dz1 = torch.empty(lib.gnbv_encoder_y1_elems(batch, grid), dtype=torch.float32, device=dev)
        ps = (seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias, seq[3].weight, seq[3].bias, seq[4].weight, seq[4].bias)
        # write-through (ops/direct_grad.py): the kernels store into the parameters' .grad slices
        direct = bool(ctx.write_through) and all(t.grad is not None and t.grad.is_contiguous() for t in ps)
        grads = [t.grad if direct else torch.empty_like(t) for t in ps]
        gs = _lib.GnbvEncoderGrads()
        for name, t in zip(("w1", "b1", "bn1_w", "bn1_b", "w2", "b2", "bn2_w", "bn2_b"), grads):
            setattr(gs, name, t.data_ptr())
        params = _params_struct(seq, ctx.grid_i8, ctx.autocorr, ctx.dp, ctx.guard)
        ws = _workspace(lib, batch, grid, dev)
        _lib.check(lib.gnbv_encoder_grid_backward(
            ctx.obs_ptr, _lib.ptr(rows), base.stride(0), batch, grid, C.byref(params), y1.data_ptr(),
            y2.data_ptr(), bn_state.data_ptr(), d_feats.data_ptr(), dy2.data_ptr(), dz1.data_ptr(), C.byref(gs),
            ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)), "gnbv_encoder_grid_backward")
        if direct:
            return (None,) * 22
        return (None,) * 14 + tuple(grads)


def grid_encoder(base: torch.Tensor, rows: Optional[torch.Tensor], grid_off: int, grid: int, seq, training: bool,
                 skip_flag: Optional[torch.Tensor] = None, write_through: bool = False,
                 grid_i8: Optional[torch.Tensor] = None, compact: bool = False, autocorr: Optional[torch.Tensor] = None, dp=None,
                 guard=None, fold: bool = False):
    """seq = the `naive_encoder_grid` nn.Sequential (conv, bn, relu, conv, bn, relu).  `compact`: `base` rows carry
    no grid slice, the grid is read from `grid_i8` only.  `autocorr`: per-row input autocorrelation (input_autocorr).
    `guard` = (force_fp32, range_flag int32 [1] or None[, autocorr_total int32 [768] or None]): GnbvEncoderParams.force_fp32 /
    .range_flag / .autocorr_total.  `fold`: returns (y2 [B, 16 P2] -- the second conv's raw output --, bn_state) for
    linear_relu(y2, lin, fold=(bn_state, P2, range_flag)) instead of the features (include/gennbv_hip.h gnbv_linear_forward_fold)."""
    return _run(_GridEncoderFn, base, rows, grid_off, grid, training, skip_flag, seq, bool(write_through), grid_i8, bool(compact), autocorr, dp, guard, bool(fold), seq[0].weight, seq[0].bias,
                                seq[1].weight, seq[1].bias, seq[3].weight, seq[3].bias, seq[4].weight, seq[4].bias)


class _LinearReluFn(torch.autograd.Function):
    """relu(x @ w.T + b) on the split-K MFMA kernel (csrc/linear.hip); backward = three library GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b, mod=None, fp32_arith=False, bn_state=None, fold_p=0, range_flag=None):
        ctx.mod = mod  # write-through target (ops/direct_grad.py) or None
        # bn_state given: x is a BatchNorm pre-activation [M][C][fold_p] and the layer's input is relu(scale[c] x + shift[c]),
        # formed in the kernels' operand loads (scale | shift = floats 64..96 of bn_state); the input gradient returned by
        # backward is the gradient of THAT input (the producer, _GridEncoderFn, runs BatchNorm's backward from it)
        ctx.fold_p = int(fold_p) if bn_state is not None else 0
        ctx.fp32_arith = bool(fp32_arith)  # operand outside the split-f16 ranges: the fp32-MFMA kernel forward, library GEMMs backward
        lib = _lib.load()
        _lib.require_cuda(x, w, b)
        x = x.contiguous()
        m, k = x.shape
        n = w.shape[0]
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
        # one workspace per LAYER (keyed by its weight storage), not per shape: same-shape layers of another model / policy
        # instance, or of the pose branch on the second stream, must not share split-K partials
        key = ("lin", m, n, k, str(x.device), w.data_ptr())
        ws = _ws_cache.get(key)
        if ws is None:
            ws = torch.empty(lib.gnbv_linear_workspace_bytes(m, n, k), dtype=torch.uint8, device=x.device)
            _ws_cache[key] = ws
        if ctx.fold_p:
            assert not fp32_arith
            sc = bn_state.data_ptr() + 4 * 64
            _lib.check(lib.gnbv_linear_forward_fold(x.data_ptr(), sc, sc + 4 * 16, ctx.fold_p, _lib.ptr(range_flag), w.data_ptr(), b.data_ptr(), m, n, k, 1,
                                                    out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)), "gnbv_linear_forward_fold")
            ctx.save_for_backward(x, w, out, bn_state)
            return out
        _lib.check(lib.gnbv_linear_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), m, n, k, 1 | (2 if fp32_arith else 0), out.data_ptr(), ws.data_ptr(),
                                           ws.numel(), _lib.stream_ptr(x.device)), "gnbv_linear_forward")
        ctx.save_for_backward(x, w, out)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, w, out = ctx.saved_tensors[:3]
        bn_state = ctx.saved_tensors[3] if ctx.fold_p else None
        mod = ctx.mod
        direct = mod is not None and mod.weight.grad is not None and mod.bias.grad is not None
        defer = direct and getattr(mod, "_async_wgrad", False) and getattr(mod, "_defer_wgrad", False)
        m, k = x.shape
        n = w.shape[0]
        if (os.environ.get("GENNBV_CONV_SPLIT", "1") != "0" and not ctx.fp32_arith and m % 16 == 0 and m <= 128 and n % 16 == 0 and n <= 256 and k % 4 == 0 and k >= 64
                and x.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()):
            # hand-written split-f16 products (csrc/linear.hip): prep (mask, row scales, operand images, db), then dx here and
            # dW either here or -- deferred, see below -- on the second stream
            lib = _lib.load()
            d_out = d_out.contiguous()
            dev = x.device
            # per layer as well: a deferred dW launch (below) reads the operand images prep leaves here long after this
            # function returns, so another layer of the same shape must never write the same buffer in between
            key = ("linbwd", m, n, k, str(dev), w.data_ptr())
            ws = _ws_cache.get(key)
            if ws is None:
                ws = torch.empty(lib.gnbv_linear_bwd_workspace_bytes(m, n, k), dtype=torch.uint8, device=dev)
                _ws_cache[key] = ws
            if any(ws is t for _, keep, _ in _deferred_wgrad for t in keep):
                raise RuntimeError("linear_relu backward: this layer's previous deferred weight gradient has not been joined "
                                   "(call join_async_wgrads() after every backward that sets _defer_wgrad)")
            db = mod.bias.grad if direct else torch.empty(n, dtype=torch.float32, device=dev)
            dw = mod.weight.grad if direct else torch.empty(n, k, dtype=torch.float32, device=dev)
            _lib.check(lib.gnbv_linear_bwd_prep(d_out.data_ptr(), out.data_ptr(), m, n, db.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)),
                       "gnbv_linear_bwd_prep")
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty(m, k, dtype=torch.float32, device=dev)
                _lib.check(lib.gnbv_linear_bwd_dx(ws.data_ptr(), w.data_ptr(), m, n, k, dx.data_ptr(), _lib.stream_ptr(dev)), "gnbv_linear_bwd_dx")

            # (write-through layers whose owner asked for it: sum(dW^2) leaves the GEMM as fp64 partial sums, the optimizer's norm
            # pass then skips this gradient -- ops/ppo_ops.py FlatAdam.step(sq_slice=...))
            sq = getattr(mod, "_dw_sq_partial", None) if direct else None
            if mod is not None:
                mod._dw_sq_written = sq is not None

            def launch_dw():
                if bn_state is not None:
                    sc = bn_state.data_ptr() + 4 * 64
                    _lib.check(lib.gnbv_linear_bwd_dw_fold(ws.data_ptr(), x.data_ptr(), sc, sc + 4 * 16, ctx.fold_p, m, n, k, dw.data_ptr(), _lib.ptr(sq),
                                                           _lib.stream_ptr(dev)), "gnbv_linear_bwd_dw_fold")
                    return
                if sq is not None:
                    _lib.check(lib.gnbv_linear_bwd_dw_sq(ws.data_ptr(), x.data_ptr(), m, n, k, dw.data_ptr(), sq.data_ptr(), _lib.stream_ptr(dev)),
                               "gnbv_linear_bwd_dw_sq")
                    return
                _lib.check(lib.gnbv_linear_bwd_dw(ws.data_ptr(), x.data_ptr(), m, n, k, dw.data_ptr(), _lib.stream_ptr(dev)), "gnbv_linear_bwd_dw")
            if defer:
                evt = torch.cuda.Event()
                evt.record(torch.cuda.current_stream(dev))  # (dW needs prep's images only, not the dx product)
                _deferred_wgrad.append((launch_dw, (x, ws) if bn_state is None else (x, ws, bn_state), evt))
            else:
                launch_dw()
            return (dx, None, None) + (None,) * 6 if direct else (dx, dw, db) + (None,) * 6
        if mod is not None:
            mod._dw_sq_written = False
        if bn_state is not None:  # (shapes outside the hand-written backward: the activations once, in torch)
            c = k // ctx.fold_p
            x = torch.relu(x.view(m, c, ctx.fold_p) * bn_state[64:64 + c].view(1, c, 1) + bn_state[64 + 16:64 + 16 + c].view(1, c, 1)).view(m, k)
        g = torch.ops.aten.threshold_backward(d_out.contiguous(), out, 0.0)
        if defer:
            # Nothing downstream of this node needs dW / db (only the optimizer does): they are computed on a second stream
            # beside the rest of the backward.  The launch itself is DEFERRED to join_async_wgrads(), i.e. until the main
            # backward has been issued: under hipGraph replay the executor keeps a node on the queue of the first child captured
            # after its parent, so launching the dW GEMM here moved the conv-stack backward -- the critical path -- to another
            # queue (a ~9 us hand-over) and made dW wait for the dx GEMM it does not depend on.  It needs g only: the event.
            # (`_defer_wgrad` is raised by the caller around its backward() only: nobody else would call the join.)
            evt = torch.cuda.Event()
            evt.record(torch.cuda.current_stream(g.device))

            def launch_lib():
                torch.mm(g.t(), x, out=mod.weight.grad)
                torch.sum(g, 0, out=mod.bias.grad)
            _deferred_wgrad.append((launch_lib, (g, x), evt))
            dx = g @ w if ctx.needs_input_grad[0] else None
            return (dx, None, None) + (None,) * 6
        dx = g @ w if ctx.needs_input_grad[0] else None
        if direct:
            _wide_dw(g, x, mod.weight.grad)
            torch.sum(g, 0, out=mod.bias.grad)
            return (dx, None, None) + (None,) * 6
        return (dx, _wide_dw(g, x, None), g.sum(0)) + (None,) * 6


def _wide_dw(g: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
    """dW = g^T x for a layer applied to MANY rows (the semantic branch's patch embedding: 8192 rows, 64 x 128 weights): the
    contraction runs over the rows, which is the split-K linear kernel's shape -- out[n][k] = sum_m g^T[n][m] x^T[k][m] -- in its fp32-MFMA
    flavour (gradients have no fixed range for the split-f16 scalings).  The library picked a 60 us kernel for this 0.13 GFLOP product,
    on the update's critical path (profiles/r04_semantic_minibatch_timeline.txt)."""
    m, n = g.shape
    k = x.shape[1]
    if not (g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and m >= 1024 and m % 8 == 0 and k % 64 == 0 and n <= 128):
        if out is None:
            return g.t() @ x
        return torch.mm(g.t(), x, out=out)
    lib = _lib.load()
    dev = g.device
    gt, xt = g.t().contiguous(), x.t().contiguous()  # [n][m], [k][m]
    key = ("wide_dw", n, k, m, str(dev))
    cached = _ws_cache.get(key)
    if cached is None:
        cached = (torch.empty(lib.gnbv_linear_workspace_bytes(n, k, m), dtype=torch.uint8, device=dev), torch.zeros(k, dtype=torch.float32, device=dev))
        _ws_cache[key] = cached
    ws, zero_bias = cached
    dst = out
    if dst is None or not dst.is_contiguous() or dst.data_ptr() % 16:  # (a slice of the flat gradient buffer need not be 16-byte aligned)
        dst = torch.empty(n, k, dtype=torch.float32, device=dev)
    _lib.check(lib.gnbv_linear_forward(gt.data_ptr(), xt.data_ptr(), zero_bias.data_ptr(), n, k, m, 2, dst.data_ptr(), ws.data_ptr(), ws.numel(),
                                       _lib.stream_ptr(dev)), "gnbv_linear_forward (wide dW)")
    if out is None:
        return dst
    if dst is not out:
        out.copy_(dst)
    return out


_deferred_wgrad = []


def join_async_wgrads(device, join: bool = True) -> None:
    """Launch the deferred weight-gradient GEMMs (`_async_wgrad`) on the second stream and make the current stream wait for them
    (`join=False`: the caller orders its consumers behind `second_stream(device)` itself -- the data-parallel step's exchange)."""
    if not _deferred_wgrad:
        return
    cur = torch.cuda.current_stream(device)
    side = _side_stream(device, 0)  # the pose branch's stream: a third stream makes the graph scheduler serialise branches
    while _deferred_wgrad:
        launch, keep, evt = _deferred_wgrad.pop()
        side.wait_event(evt)
        with torch.cuda.stream(side), torch.no_grad():
            launch()
        for t in keep:
            t.record_stream(side)
    if join:
        cur.wait_stream(side)


def linear_relu(x: torch.Tensor, lin: torch.nn.Linear, fold=None) -> torch.Tensor:
    """Linear + ReLU of the K-dominated fc layer (hybrid_encoder.py:39-42 of the reference).  fold = (bn_state, P, range_flag) from
    grid_encoder(..., fold=True) after linear_fold_ok(...): x is the conv stack's raw output, BatchNorm-2 + ReLU happen in the operand load."""
    n, k = lin.weight.shape
    if fold is not None:
        return _run(_LinearReluFn, x, lin.weight, lin.bias, lin if getattr(lin, "_grad_write_through", False) else None, False, fold[0], fold[1], fold[2])
    if k % 4 or n % 64 or not lin.weight.is_contiguous() or x.dtype != torch.float32:
        return torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias))
    return _run(_LinearReluFn, x, lin.weight, lin.bias, lin if getattr(lin, "_grad_write_through", False) else None,
                               bool(getattr(lin, "_fp32_arith", False)))


def linear_fold_ok(lin: torch.nn.Linear, m: int, p: int, force_fp32: bool) -> bool:
    """Can fc layer `lin` take its input as (pre-BatchNorm activations, scale, shift) -- linear_relu(..., fold=...)?"""
    n, k = lin.weight.shape
    if force_fp32 or getattr(lin, "_fp32_arith", False) or getattr(lin, "_no_fold", False):  # (_no_fold: tests, the materialised-feature path)
        return False
    if n % 64 or not lin.weight.is_contiguous() or lin.weight.dtype != torch.float32:
        return False
    return bool(_lib.load().gnbv_linear_fold_ok(int(m), int(n), int(k), int(p)))


def hybrid_forward(enc, observations) -> torch.Tensor:
    """Hybrid_Encoder.forward with the grid branch on the gfx950 kernels."""
    feature_action, feature_grid = hybrid_branches(enc, observations)  # (feature_grid = [grid | semantic] with the opt-in branch)
    return enc.output_layer(torch.cat((feature_action, feature_grid), dim=-1))


def semantic_features(enc, observations) -> torch.Tensor:
    """The opt-in semantic branch (network/hybrid_encoder.py): two 64 x 64 gray frames -> [B, 256] on the linear kernels."""
    s, g = enc.state_input_shape[0], enc.grid_size
    if isinstance(observations, (RowGather, DenseObs)):
        off = s if observations.compact_state_dim is not None else s + g ** 3
        rgb = observations.base[:, off:off + 8192]
        if observations.rows is not None:
            rgb = rgb[observations.rows]
    else:
        rgb = observations[:, s + g ** 3:s + g ** 3 + 8192]
    emb = linear_relu(enc.rgb_patches(rgb.float()).contiguous(), enc.naive_encoder_rgb[0])
    return linear_relu(emb.reshape(rgb.shape[0], -1), enc.output_layer_rgb[0])


def hybrid_branches(enc, observations):
    """(pose-history features, grid features), each [B, 256]: Hybrid_Encoder.forward up to the concat
    (hybrid_encoder.py:76-88 of the reference)."""
    s = enc.state_input_shape[0]
    g = enc.grid_size
    grid_i8, compact, autocorr = None, False, None
    if isinstance(observations, (RowGather, DenseObs)):
        base, rows, grid_i8, autocorr = observations.base, observations.rows, observations.grid_i8, observations.autocorr
        compact = observations.compact_state_dim is not None
        num_env = int(rows.shape[0]) if rows is not None else int(base.shape[0])
        get_state = lambda: observations.columns(0, s)  # noqa: E731  (gather of the pose columns: on the side stream too)
    else:
        _lib.require_cuda(observations)  # no CPU implementation in the product (tests/torch_reference.py is the checker)
        base, rows = observations, None
        num_env = int(observations.shape[0])
        get_state = lambda: observations[:, :s]  # noqa: E731
        if base.stride(1) != 1:
            base = base.contiguous()

    def pose_branch():
        if s % 6 == 0 and base.dtype == torch.float32 and base.stride(1) == 1:
            # gather + positional encoding in one launch (csrc/linear.hip k_pose_encode)
            lib = _lib.load()
            action_input = torch.empty(num_env, 4 * s, dtype=torch.float32, device=base.device)
            _lib.check(lib.gnbv_pose_encode(base.data_ptr(), _lib.ptr(rows), base.stride(0), num_env, s // 6, action_input.data_ptr(),
                                            _lib.stream_ptr(base.device)), "gnbv_pose_encode")
        else:
            state = get_state()
            action_input = enc.positional_encoding(state.view(num_env, -1, 6)).view(num_env, -1)
        seq = enc.naive_encoder_action  # Linear, ReLU, Linear, ReLU (hybrid_encoder.py:43-45 of the reference)
        if len(seq) == 4 and isinstance(seq[0], torch.nn.Linear) and isinstance(seq[2], torch.nn.Linear):
            # the same split-K / skinny-GEMM kernels as fc_grid (csrc/linear.hip): the library picked 41 + 44 us kernels for these
            # two small products, which ran beside -- and slowed -- the conv kernels
            return linear_relu(linear_relu(action_input, seq[0]), seq[2])
        return seq(action_input)

    # The pose-history branch (a gather, a few small GEMMs and element-wise kernels) is independent of the
    # grid branch until the concat: fork it onto a second HIP stream so that it overlaps the conv kernels
    # in the forward AND (autograd replays each op on the stream it was recorded on) in the backward.
    side = _side_stream(base.device) if enc.overlap_branches else None
    if side is not None:
        cur = torch.cuda.current_stream(base.device)
        side.wait_stream(cur)  # the fork point; the pose kernels themselves are launched AFTER the grid branch (below)
    else:
        feature_action = pose_branch()
        feature_sem = semantic_features(enc, observations) if getattr(enc, "semantic_branch", False) else None
    p2 = conv_out(conv_out(g)) ** 3
    fold = linear_fold_ok(enc.output_layer_grid[0], num_env, p2, getattr(enc, "force_fp32", False))
    feature_grid = grid_encoder(base, rows, s, g, enc.naive_encoder_grid, enc.training, getattr(enc, "_bn_skip_flag", None),
                                getattr(enc, "_grad_write_through", False), grid_i8, compact, autocorr,
                                getattr(enc, "_dp_sync", None),
                                (getattr(enc, "force_fp32", False), getattr(enc, "_range_flag", None),
                                 getattr(enc, "_autocorr_total", None) if (enc.training and autocorr is not None) else None), fold)
    fold_args = None
    if fold:  # (y2, bn_state): BatchNorm-2 + ReLU are formed inside fc_grid's kernels
        feature_grid, bn_state = feature_grid
        fold_args = (bn_state, p2, getattr(enc, "_range_flag", None))
    if getattr(enc, "_split_backward", False) and torch.is_grad_enabled():
        # data-parallel: cut the autograd graph at the conv-stack output so that the backward runs in
        # two phases (late layers first, their gradient all-reduce overlaps the conv-stack backward)
        enc._grid_feats_out = feature_grid
        feature_grid = feature_grid.detach().requires_grad_(True)
        enc._grid_feats_leaf = feature_grid
    feature_grid = linear_relu(feature_grid, enc.output_layer_grid[0], fold_args)  # Linear + ReLU, split-K MFMA kernel
    if side is not None:
        # Launch order matters under hipGraph replay: the executor keeps a node on the queue of the FIRST child captured after
        # its parent, so with the pose branch captured first the conv chain -- the critical path -- was moved to a second queue and
        # paid a ~10 us cross-queue hand-over at the fork and again at the join (profiles/r02_notes.md).
        with torch.cuda.stream(side):
            feature_action = pose_branch()
            feature_sem = semantic_features(enc, observations) if getattr(enc, "semantic_branch", False) else None
        torch.cuda.current_stream(base.device).wait_stream(side)
        feature_action.record_stream(torch.cuda.current_stream(base.device))
        if feature_sem is not None:
            feature_sem.record_stream(torch.cuda.current_stream(base.device))
        if getattr(enc, "_defer_pose_backward", False) and torch.is_grad_enabled() and feature_action.requires_grad:
            # The same rule for the backward: autograd would run (capture) the pose branch's backward BEFORE the grid branch's
            # (its nodes were created later).  Cut the graph at the branch output; the owner of the flag runs
            # pose_branch_backward() after the main backward, on the second stream, from an event recorded when the head's
            # backward produced this leaf's gradient.
            leaf = feature_action.detach().requires_grad_(True)
            evt = torch.cuda.Event()
            leaf.register_hook(lambda g, e=evt, d=base.device: e.record(torch.cuda.current_stream(d)))
            enc._pose_deferred = (feature_action, leaf, evt)
            feature_action = leaf
    if feature_sem is not None:  # output_layer's columns: [pose | grid | semantic]
        feature_grid = torch.cat((feature_grid, feature_sem), dim=-1)
    return feature_action, feature_grid


def second_stream(device):
    """The stream the pose branch and the deferred weight-gradient GEMMs run on."""
    return _side_stream(device, 0)


def pose_branch_backward(enc, device, join: bool = True) -> None:
    """Second half of the `_defer_pose_backward` protocol (see hybrid_branches): the pose branch's backward on the second
    stream, joined into the current stream (unless `join=False`, see join_async_wgrads)."""
    pending = getattr(enc, "_pose_deferred", None)
    if pending is None:
        return
    enc._pose_deferred = None
    out, leaf, evt = pending
    if leaf.grad is None:
        return
    side, cur = _side_stream(device), torch.cuda.current_stream(device)
    side.wait_event(evt)
    # (the gradient was allocated on the current stream and is dropped when this function returns: without the join below the caching
    # allocator would hand its block to the next allocation on the current stream while the second stream still reads it -- seen as wrong
    # updates in the replayed data-parallel graph, where no host time hides the race)
    if not join:
        leaf.grad.record_stream(side)
    with torch.cuda.stream(side):
        torch.autograd.backward([out], [leaf.grad])
    if join:
        cur.wait_stream(side)


class _PolicyHeadFn(torch.autograd.Function):
    """output_layer (Linear + ReLU over the implicit concat) + action_net + value_net, fused
    (csrc/head.hip): 2 launches forward, 2 backward."""

    @staticmethod
    def forward(ctx, fa, fg, w_out, b_out, w_act, b_act, w_val, b_val, mods):
        lib = _lib.load()
        _lib.require_cuda(fa, fg, w_out)
        fa, fg = fa.contiguous(), fg.contiguous()
        m, k1 = fa.shape
        k2, f, a = fg.shape[1], w_out.shape[0], w_act.shape[0]
        dev = fa.device
        feat = torch.empty(m, f, dtype=torch.float32, device=dev)
        logits = torch.empty(m, a, dtype=torch.float32, device=dev)
        values = torch.empty(m, dtype=torch.float32, device=dev)
        _lib.check(lib.gnbv_policy_head_forward(fa.data_ptr(), fg.data_ptr(), m, k1, k2, w_out.data_ptr(), b_out.data_ptr(), f,
                                                w_act.data_ptr(), b_act.data_ptr(), a, w_val.data_ptr(), b_val.data_ptr(),
                                                feat.data_ptr(), logits.data_ptr(), values.data_ptr(), _lib.stream_ptr(dev)),
                   "gnbv_policy_head_forward")
        ctx.save_for_backward(fa, fg, feat, w_out, w_act, w_val)
        ctx.mods = mods
        ctx.mark_non_differentiable(feat)
        ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for `feat`
        return logits, values, feat

    @staticmethod
    def backward(ctx, d_logits, d_values, _d_feat):
        lib = _lib.load()
        fa, fg, feat, w_out, w_act, w_val = ctx.saved_tensors
        m, k1 = fa.shape
        k2, f, a = fg.shape[1], w_out.shape[0], w_act.shape[0]
        dev = fa.device
        d_logits, d_values = d_logits.contiguous(), d_values.contiguous()
        d_fa, d_fg = torch.empty_like(fa), torch.empty_like(fg)
        dh = torch.empty(m, f, dtype=torch.float32, device=dev)
        ps = None
        if ctx.mods is not None:  # write-through (ops/direct_grad.py)
            lo, la, lv = ctx.mods
            ps = [lo.weight.grad, lo.bias.grad, la.weight.grad, la.bias.grad, lv.weight.grad, lv.bias.grad]
            if any(g is None or not g.is_contiguous() for g in ps):
                ps = None
        direct = ps is not None
        if not direct:
            ps = [torch.empty_like(w_out), torch.empty(f, device=dev), torch.empty_like(w_act), torch.empty(a, device=dev),
                  torch.empty_like(w_val), torch.empty(1, device=dev)]
        _lib.check(lib.gnbv_policy_head_backward(fa.data_ptr(), fg.data_ptr(), m, k1, k2, feat.data_ptr(), d_logits.data_ptr(),
                                                 d_values.data_ptr(), w_out.data_ptr(), f, w_act.data_ptr(), a, w_val.data_ptr(),
                                                 dh.data_ptr(), d_fa.data_ptr(), d_fg.data_ptr(), *[t.data_ptr() for t in ps],
                                                 _lib.stream_ptr(dev)), "gnbv_policy_head_backward")
        if direct:
            return d_fa, d_fg, None, None, None, None, None, None, None
        return (d_fa, d_fg, *ps, None)


def policy_head_supported(enc, action_net, value_net) -> bool:
    lo = enc.output_layer[0]
    return (isinstance(action_net, torch.nn.Linear) and isinstance(value_net, torch.nn.Linear) and value_net.out_features == 1
            and lo.in_features % 32 == 0 and lo.out_features % 16 == 0 and action_net.in_features == lo.out_features
            and value_net.in_features == lo.out_features and lo.weight.dtype == torch.float32)


def policy_head(enc, action_net, value_net, feature_action, feature_grid):
    """(logits [B, A], values [B], features [B, F]) from the two encoder branches."""
    lo = enc.output_layer[0]
    wt = all(getattr(m, "_grad_write_through", False) for m in (lo, action_net, value_net))
    return _run(_PolicyHeadFn, feature_action, feature_grid, lo.weight, lo.bias, action_net.weight, action_net.bias, value_net.weight,
                               value_net.bias, (lo, action_net, value_net) if wt else None)


_side_streams = {}


def _side_stream(device, which: int = 0):
    key = (str(device), which)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device)
    return _side_streams[key]
