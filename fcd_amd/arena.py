"""Flat parameter / gradient arenas.

All parameters of a network live in ONE contiguous fp32 buffer and their gradients in
another, each parameter at a 256-byte aligned offset (``p.data`` and ``p.grad`` become views).
That buys three things on the hot path:

* inside ``with arena.accumulating():`` the network's backward writes weight gradients straight
  into ``p.grad`` (the wgrad / column-sum kernels accumulate in place), so autograd's
  per-parameter ``add`` launches and the temporaries behind them disappear.  It is opt-in per
  backward call: ``torch.autograd.grad(y, x, create_graph=True)`` (the reference's
  ``gradient_penalty``, ``fcd_solver.py:230-235``) must not touch ``p.grad``, so outside the
  context gradients travel through autograd as usual;
* Adam is one launch over the flat buffers instead of one per parameter;
* data-parallel training all-reduces the gradient arena itself (RCCL, a few large chunks),
  no pack / unpack copies.

Padding elements have zero gradient for ever, so they stay zero under Adam.
"""
from __future__ import annotations

import contextlib
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
                p._fcd_arena = self      # grad_sink() finds the arena (and its `direct` switch) here
        self.direct = False              # True only inside accumulating()
        self.index = {id(p): i for i, p in enumerate(self.params)}

    @contextlib.contextmanager
    def accumulating(self):
        """Backward passes started inside this context accumulate parameter gradients straight into
        ``p.grad`` (the arena) and hand autograd ``None`` for them.  A plain attribute, not a
        thread-local: autograd runs the backward nodes on its own worker thread."""
        prev, self.direct = self.direct, True
        try:
            yield self
        finally:
            self.direct = prev

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
            p._fcd_arena = None

    def zero_grad(self) -> None:
        self.grad.zero_()

    def bump_versions(self) -> None:
        """The flat buffer was updated in place by a raw kernel: tell autograd / the packed-weight
        caches that every parameter changed."""
        for p in self.params:
            torch.autograd.graph.increment_version(p)


def grad_sink(p: torch.Tensor, need: bool):
    """The tensor a backward kernel should accumulate this parameter's gradient into, or None
    when autograd has to carry it (no arena, not inside ``arena.accumulating()``, or the gradient
    is not wanted)."""
    ar = getattr(p, "_fcd_arena", None)
    if need and ar is not None and ar.direct and p.grad is not None:
        return p.grad
    return None
