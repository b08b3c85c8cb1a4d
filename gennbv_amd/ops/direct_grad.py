"""Gradient write-through for the PPO update (no AccumulateGrad kernels).

In the hipGraph-captured minibatch step every parameter receives exactly ONE gradient contribution per
step and `FlatAdam` has already pointed each `.grad` at its slice of the flat gradient buffer.  The stock
autograd path still materialises every weight gradient in a fresh tensor and then ADDS it into `.grad`
(one more element-wise kernel per parameter, ~20 per minibatch; 25 us for the 55 MB fc_grid weight alone)
after a zero-fill.  With write-through the backward GEMM writes `dW` (and the bias reduction `db`)
straight into the slice (`out=`) and returns no gradient for the parameter.

Only valid while (a) `.grad` exists, (b) nobody relies on accumulation across backward passes: it is
switched on by `PPO_Grid_Obs._hip_setup` for the training path and off everywhere else.
"""
from __future__ import annotations

import types

import torch


class _LinearWT(torch.autograd.Function):
    # weight / bias are passed as tensor inputs so that the node exists even when x needs no gradient
    # (first layer of a branch); their gradients are written through, so backward returns None for them.
    @staticmethod
    def forward(ctx, x, weight, bias, mod):
        ctx.mod = mod
        ctx.save_for_backward(x)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        g = g.contiguous()
        dx = g @ mod.weight if ctx.needs_input_grad[0] else None
        torch.mm(g.t(), x, out=mod.weight.grad)
        torch.sum(g, 0, out=mod.bias.grad)
        return dx, None, None, None


def _forward(self, x):
    if (torch.is_grad_enabled() and getattr(self, "_grad_write_through", False) and self.bias is not None
            and self.weight.grad is not None and self.bias.grad is not None and self.weight.requires_grad):
        if x.dim() == 2:
            return _LinearWT.apply(x, self.weight, self.bias, self)
        return _LinearWT.apply(x.reshape(-1, x.shape[-1]), self.weight, self.bias, self).view(*x.shape[:-1], self.weight.shape[0])
    return torch.nn.functional.linear(x, self.weight, self.bias)


def enable(module: torch.nn.Module, on: bool = True) -> int:
    """Switch gradient write-through on/off for every nn.Linear below `module`; returns their number."""
    n = 0
    for m in module.modules():
        if isinstance(m, torch.nn.Linear):
            if on and not getattr(m, "_wt_patched", False):
                m.forward = types.MethodType(_forward, m)
                m._wt_patched = True
            m._grad_write_through = bool(on)
            n += 1
        if hasattr(m, "naive_encoder_grid"):
            m._grad_write_through = bool(on)  # conv stack: encoder_ops._GridEncoderFn writes into .grad
    return n
