"""Loss terms of the FCD solver as autograd nodes over the HIP reductions.

Each forward is a fixed-order two-stage reduction (double accumulation) that
leaves its scalar on the device; each backward is one elementwise launch that
reads the incoming scalar gradient from device memory (no host sync).
"""
from __future__ import annotations

import torch

from . import ops


class _MeanScaled(torch.autograd.Function):
    """sign * mean(x): ``-torch.mean(out_src)`` / ``torch.mean(out_src)``
    (reference ``fcd_solver.py:287, 293, 318, 324``)."""

    @staticmethod
    def forward(ctx, x, sign):
        x = x.contiguous()
        ctx.shape, ctx.scale = x.shape, sign / x.numel()
        return ops.sum_scaled(x, ctx.scale)

    @staticmethod
    def backward(ctx, g):
        return ops.fill_scaled(ctx.shape, ctx.scale, g.contiguous()), None


class _L1Mean(torch.autograd.Function):
    """``torch.mean(torch.abs(a - b))`` (reference ``fcd_solver.py:326, 354, 359``)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.l1_mean(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga, gb = ops.l1_mean_bwd(a, b, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return ga, gb


class _BCELogitsMean(torch.autograd.Function):
    """``F.binary_cross_entropy_with_logits(logit, target)`` (reference ``fcd_solver.py:263``)."""

    @staticmethod
    def forward(ctx, logit, target):
        logit, target = logit.contiguous(), target.contiguous().float()
        ctx.save_for_backward(logit, target)
        return ops.bce_logits_mean(logit, target)

    @staticmethod
    def backward(ctx, g):
        logit, target = ctx.saved_tensors
        return ops.bce_logits_mean_bwd(logit, target, g.contiguous()), None


class _GPLoss(torch.autograd.Function):
    """mean((||g_n||_2 - 1)^2) over the batch (reference ``fcd_solver.py:237-239``)."""

    @staticmethod
    def forward(ctx, g):
        g = g.contiguous()
        out, norm = ops.gp_fwd(g)
        ctx.save_for_backward(g, norm)
        return out

    @staticmethod
    def backward(ctx, gs):
        g, norm = ctx.saved_tensors
        return ops.gp_bwd(g, norm, gs.contiguous())


def mean_scaled(x, sign: float):
    return _MeanScaled.apply(x, float(sign))


def l1_mean(a, b):
    return _L1Mean.apply(a, b)


def bce_logits_mean(logit, target):
    return _BCELogitsMean.apply(logit, target)


def gp_loss(dydx):
    return _GPLoss.apply(dydx)
