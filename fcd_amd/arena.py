"""Flat parameter / gradient arenas.

All parameters of a network live in ONE contiguous fp32 buffer and their gradients in
another, each parameter at a 256-byte aligned offset (``p.data`` and ``p.grad`` become views).
That buys three things on the hot path:

* the network's backward writes weight gradients straight into ``p.grad`` (the wgrad /
  column-sum kernels accumulate in place), so autograd's per-parameter ``add`` launches and
  the temporaries behind them disappear;
* Adam is one launch over the flat buffers instead of one per parameter;
* data-parallel training all-reduces the gradient arena itself (RCCL, a few large chunks),
  no pack / unpack copies.

Padding elements have zero gradient for ever, so they stay zero under Adam.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

ALIGN = 64   # elements (256 bytes)


class ParamArena:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("ParamArena: no parameters")
        dev = self.params[0].device
        self.offsets = []
        total = 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("ParamArena: parameters must be fp32 on one device")
            self.offsets.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                v = self.flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = self.grad[off:off + p.numel()].view(p.shape)
                p._fcd_direct = True     # backward kernels accumulate into p.grad themselves

    def intact(self) -> bool:
        """False once something (``module.to``, ``zero_grad(set_to_none=True)``) replaced the views."""
        lo = self.flat.data_ptr()
        glo = self.grad.data_ptr()
        for p, off in zip(self.params, self.offsets):
            if p.data_ptr() != lo + 4 * off or p.grad is None or p.grad.data_ptr() != glo + 4 * off:
                return False
        return True

    def release(self) -> None:
        for p in self.params:
            p._fcd_direct = False

    def zero_grad(self) -> None:
        self.grad.zero_()

    def bump_versions(self) -> None:
        """The flat buffer was updated in place by a raw kernel: tell autograd / the packed-weight
        caches that every parameter changed."""
        for p in self.params:
            torch.autograd.graph.increment_version(p)


def grad_sink(p: torch.Tensor, need: bool):
    """The tensor a backward kernel should accumulate this parameter's gradient into, or None
    when autograd has to carry it (no arena, or the gradient is not wanted)."""
    if need and getattr(p, "_fcd_direct", False) and p.grad is not None:
        return p.grad
    return None
