"""Thin Python wrappers over the C ABI: allocate outputs, pass raw pointers.

Every function here launches HIP kernels from libfcd_hip.so on the current
stream.  Tensors are fp32; "nhwc" tensors are [N, H, W, CP] with CP a multiple
of 4 (pad channels are zero).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from ._lib import ConvDesc, WgradDesc, check, ptr, stream

LRELU_SLOPE = 0.01
ACT_SILU, ACT_MISH = 2.0, 3.0   # FCD_ACT_SILU / FCD_ACT_MISH (fcd_hip.h)
IN_EPS = 1e-5
IN_MOMENTUM = 0.1


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def _ws(nbytes: int, device):
    if nbytes <= 0:
        return None, 0
    b = L.workspace.get(nbytes, device)
    return b.data_ptr(), b.numel()


_SIDE = {}


class SideWork:
    """Weight gradients beside the data-gradient chain: a backward pass is a chain dz -> dgrad -> norm backward -> dz
    with one weight gradient hanging off every link; nothing in the pass reads those, so they go to a second HIP
    stream, where their bandwidth-bound tails (slab reductions) and the chain's bandwidth-bound links (norm backward)
    fill each other's idle matrix pipes.  ``run`` orders the side stream behind everything enqueued so far on the
    current one; ``join`` orders the current stream behind the side work (before the gradients are read: optimizer,
    all-reduce) and lets go of the operands kept alive for it."""

    def __init__(self, device, enabled: bool = True):
        self.side = None
        if enabled and torch.device(device).type == "cuda":
            key = torch.device(device).index
            if key not in _SIDE:
                _SIDE[key] = torch.cuda.Stream(device=device)
            self.side = _SIDE[key]
        self.keep = []
        self.dirty = False

    def run(self, fn, *operands):
        if self.side is None:
            return fn()
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            r = fn()
        self.keep.extend(operands)
        self.dirty = True
        return r

    def join(self) -> None:
        if self.dirty:
            torch.cuda.current_stream().wait_stream(self.side)
            self.dirty = False
        self.keep.clear()


class WgradPairing:
    """One weight-gradient launch per layer for the two backward passes of one ``backward()`` call that produce the same
    parameter gradient (``conv_wgrad2``): the D step's paired real / fake batch and gradient penalty, the G step's two
    passes through the generator.  Whichever pass autograd runs first leaves its operand pair per layer here
    (``leave``), the other takes it into its own launch (``take``); whatever is left when the backward call ends is
    turned into gradients by the launches the leaving pass would have made (an engine callback).  Pairs are tied to
    the autograd graph task they were left in: what an aborted call left behind is dropped with its graph."""

    def __init__(self):
        self._stash = {}          # key -> [(item, alone)]
        self._queued = False
        self._task = -1

    def empty(self) -> bool:
        return not self._stash

    def leave(self, key, item, alone) -> None:
        """``item``: what the taking pass needs (operands, gradient tensor, maxima); ``alone()``: the launch of this pair
        by itself, for the case that nobody takes it."""
        task = torch._C._current_graph_task_id()
        if task != self._task:
            self._task, self._stash, self._queued = task, {}, False
        self._stash.setdefault(key, []).append((item, alone))
        if not self._queued:
            self._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def take(self, key, match):
        """The item an earlier pass of this backward call left under ``key`` if ``match(item)``, else None."""
        if not self._stash or self._task != torch._C._current_graph_task_id():
            return None
        pend = self._stash.get(key)
        if pend and len(pend) == 1 and match(pend[0][0]):
            del self._stash[key]
            return pend[0][0]
        return None

    def pending(self, key) -> bool:
        return bool(self._stash) and key in self._stash and self._task == torch._C._current_graph_task_id()

    def drain(self, key) -> None:
        """Launch, each by itself, whatever is still left under ``key`` (a pass about to declare the gradient final)."""
        if self._stash and self._task == torch._C._current_graph_task_id():
            for _, alone in self._stash.pop(key, ()):
                alone()

    def flush(self) -> None:
        self._queued = False
        stash, self._stash = self._stash, {}
        for items in stash.values():
            for _, alone in items:
                alone()


# --------------------------------------------------------------------------- #
# convolution geometry (host logic; unit-tested on CPU)
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class ConvGeom:
    """The equivalent convolution W[O][I][k][k], stride s, padding p.

    nn.Conv2d(I, O, k, s, p) uses it directly; nn.ConvTranspose2d(O, I, k, s, p)
    (weight [O][I][k][k] in torch's [C_in, C_out, kh, kw] order) is its adjoint.
    """
    O: int
    I: int
    k: int
    s: int
    p: int

    @property
    def OP(self) -> int:
        return pad4(self.O)

    @property
    def IP(self) -> int:
        return pad4(self.I)

    def small_hw(self, bh: int, bw: int):
        """Spatial size on the conv-output ("small") side for a big-side size."""
        return (bh + 2 * self.p - self.k) // self.s + 1, (bw + 2 * self.p - self.k) // self.s + 1

    def big_hw(self, sh: int, sw: int):
        """Spatial size on the conv-input ("big") side: ConvTranspose2d output size."""
        return (sh - 1) * self.s - 2 * self.p + self.k, (sw - 1) * self.s - 2 * self.p + self.k

    def fwd_desc(self, n: int, bh: int, bw: int, precision: int = L.PREC_F32, act_slope: float = 1.0,
                 mask_slope: float = 1.0) -> ConvDesc:
        """big [n,bh,bw,IP] -> small [n,sh,sw,OP]: the conv itself."""
        sh, sw = self.small_hw(bh, bw)
        d = ConvDesc()
        d.N, d.IH, d.IW, d.Cin = n, bh, bw, self.IP
        d.GH, d.GW, d.OH, d.OW = sh, sw, sh, sw
        d.Cout, d.CoutP = self.O, self.OP
        d.KH = d.KW = self.k
        d.istride, d.ostride, d.nphase = self.s, 1, 1
        d.pad_h[0] = d.pad_w[0] = self.p
        d.splitk, d.precision = 0, precision
        d.act_slope, d.mask_slope = act_slope, mask_slope
        return d

    def adj_desc(self, n: int, sh: int, sw: int, precision: int = L.PREC_F32, act_slope: float = 1.0,
                 mask_slope: float = 1.0) -> ConvDesc:
        """small [n,sh,sw,OP] -> big [n,bh,bw,IP]: the adjoint (data gradient / transposed conv)."""
        bh, bw = self.big_hw(sh, sw)
        d = ConvDesc()
        d.N, d.IH, d.IW, d.Cin = n, sh, sw, self.OP
        d.OH, d.OW, d.Cout, d.CoutP = bh, bw, self.I, self.IP
        d.istride = 1
        if self.s == 1:
            d.GH, d.GW = bh, bw
            d.KH = d.KW = self.k
            d.ostride, d.nphase = 1, 1
            d.pad_h[0] = d.pad_w[0] = self.k - 1 - self.p
        elif self.s == 2 and self.k == 4 and self.p == 1:
            d.GH, d.GW = sh, sw
            d.KH = d.KW = 2
            d.ostride, d.nphase = 2, 4
            for ph in range(4):
                py, px = ph >> 1, ph & 1
                d.pad_h[ph], d.pad_w[ph] = 1 - py, 1 - px
                d.ooff_h[ph], d.ooff_w[ph] = py, px
        else:
            raise NotImplementedError(f"adjoint of conv k={self.k} s={self.s} p={self.p}")
        d.splitk, d.precision = 0, precision
        d.act_slope, d.mask_slope = act_slope, mask_slope
        return d

    def wgrad_desc(self, n: int, bh: int, bw: int, precision: int = L.PREC_F32, accumulate: bool = False) -> WgradDesc:
        sh, sw = self.small_hw(bh, bw)
        d = WgradDesc()
        d.N, d.GH, d.GW, d.O, d.OP = n, sh, sw, self.O, self.OP
        d.BH, d.BW, d.I, d.IP = bh, bw, self.I, self.IP
        d.KH = d.KW = self.k
        d.stride, d.pad = self.s, self.p
        d.splitk, d.precision, d.accumulate = 0, precision, int(accumulate)
        return d

    def adj_phases(self) -> int:
        return 1 if self.s == 1 else 4

    def adj_taps(self) -> int:
        return self.k if self.s == 1 else 2


def weight_format(d: ConvDesc) -> int:
    """0 = fp32 pack, 1 = three bf16 planes, 2 = two fp16 planes of the scaled weights (PREC_F16X2), 3 = fp32 pack and
    the kernel takes the two-fp16-piece form if both maxima are set (PREC_F16X2 on the 7x7 halo kernels)."""
    return int(L.load().fcd_conv_weight_format(C.byref(d)))


_OVERFLOW = {}     # device index -> int32 word registered with fcd_set_overflow_flag


def overflow_flag(device) -> torch.Tensor:
    """The device's sticky overflow word of PREC_F16X2 (``fcd_set_overflow_flag``): every fp32x2 kernel and weight
    pack stores 1 to it when a scaled operand exceeds 2^15 (a stale or under-estimated max |.|).
    Created, zeroed and registered on first use; it lives as long as the process."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    t = _OVERFLOW.get(idx)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
        L.check(L.load().fcd_set_overflow_flag(idx, t.data_ptr()), "fcd_set_overflow_flag")
        _OVERFLOW[idx] = t
    return t


def check_overflow(device, reset: bool = True) -> None:
    """Poll the overflow word (one device-to-host copy: a synchronisation point) and raise if it is set."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    t = _OVERFLOW.get(idx)
    if t is None:
        return
    if int(t.item()) != 0:
        if reset:
            t.zero_()
        raise RuntimeError("fcd_amd: fp32x2 overflow guard: a scaled operand exceeded 2^15 in a kernel on "
                           f"cuda:{idx} -- the max |.| attached to a descriptor (a_amax / w_amax / g_amax / x_amax) was "
                           "stale or too small for the tensor it describes; results since the last check may hold Inf / NaN")


def _amax_ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() < AMAX_SLOTS or t.data_ptr() % 16:
        raise ValueError("an amax operand is a contiguous float32 device tensor of AMAX_SLOTS (1024) partial maxima, 16-byte aligned")
    overflow_flag(t.device)      # fp32x2 in use on this device: make sure its overflow word is registered
    return t.data_ptr()


def set_amax(d: ConvDesc, a_amax: Optional[torch.Tensor], w_amax: Optional[torch.Tensor]) -> ConvDesc:
    """Attach the device scalars max |activation operand| / max |weight| that PREC_F16X2 scales its operands by."""
    d.a_amax = _amax_ptr(a_amax)
    d.w_amax = _amax_ptr(w_amax)
    d._amax_refs = (a_amax, w_amax)        # the descriptor holds raw addresses: keep the tensors alive with it
    return d


def set_out_amax(d: ConvDesc, out_amax: Optional[torch.Tensor]) -> ConvDesc:
    """Ask ``conv_gather`` to leave max |out| in ``out_amax`` (zeroed AMAX_SLOTS floats): the next layer's ``a_amax``."""
    d.out_amax = _amax_ptr(out_amax)
    d._out_amax_ref = out_amax
    return d


def set_wgrad_amax(d: WgradDesc, g_amax: Optional[torch.Tensor], x_amax: Optional[torch.Tensor]) -> WgradDesc:
    """PREC_F16X2 weight gradients: device scalars max |gsmall| / max |xbig| (used where the 3x3 halo kernel applies)."""
    d.g_amax = _amax_ptr(g_amax)
    d.x_amax = _amax_ptr(x_amax)
    d._amax_refs = (g_amax, x_amax)
    return d


def _check_packed(d: ConvDesc, w_packed: torch.Tensor, K: int) -> None:
    """fp32 pack, or the three-plane bf16 / two-plane fp16 pack where the descriptor asks for it."""
    fmt = weight_format(d)
    if fmt == 2:
        L.require_gpu_tensor(w_packed, "w_packed (h2)", torch.float16)
        if not d.a_amax or not d.w_amax:
            raise RuntimeError("conv: PREC_F16X2 needs ops.set_amax(d, a_amax, w_amax)")
        want = 2 * d.nphase * d.Cout * K
    elif fmt == 1:
        L.require_gpu_tensor(w_packed, "w_packed (x3)", torch.bfloat16)
        want = 3 * d.nphase * d.Cout * K
    else:
        L.require_gpu_tensor(w_packed, "w_packed")
        want = d.nphase * d.Cout * K
    if w_packed.numel() != want:
        raise RuntimeError(f"conv: packed weight has {w_packed.numel()} elements, expected {want}")


class KernelTimer:
    """HIP-event timing of selected conv_gather launches on the launch stream (bench.py uses it to
    time the dominant kernel inside the real training steps)."""

    def __init__(self, match):
        self.match = match      # ConvDesc -> tag (truthy) for launches to time, falsy otherwise
        self.pairs = []         # (start event, end event)
        self.info = []          # (tag, algorithmic FLOPs of the launch), aligned with pairs

    def note(self, tag, d: ConvDesc) -> None:
        self.info.append((tag, 2.0 * d.N * d.GH * d.GW * d.nphase * d.Cout * d.KH * d.KW * d.Cin))

    def mean_us(self, tag=None) -> float:
        torch.cuda.synchronize()
        sel = [p for p, (t, _) in zip(self.pairs, self.info) if tag is None or t == tag]
        return sum(a.elapsed_time(b) for a, b in sel) / max(1, len(sel)) * 1e3

    def count(self, tag=None) -> int:
        return sum(1 for t, _ in self.info if tag is None or t == tag)

    def total(self):
        """(launches, summed microseconds, summed algorithmic FLOPs) over everything timed."""
        torch.cuda.synchronize()
        us = sum(a.elapsed_time(b) for a, b in self.pairs) * 1e3
        return len(self.pairs), us, sum(f for _, f in self.info)


KERNEL_TIMER: Optional[KernelTimer] = None


def conv_gather(d: ConvDesc, x: torch.Tensor, w_packed: torch.Tensor, bias=None, mask=None, residual=None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    exp_in = (d.N, d.IH, d.IW, d.Cin)
    if tuple(x.shape) != exp_in:
        raise RuntimeError(f"conv_gather: input {tuple(x.shape)} != descriptor {exp_in}")
    K = d.KH * d.KW * d.Cin
    _check_packed(d, w_packed, K)
    oshape = (d.N, d.OH, d.OW, d.CoutP)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != oshape:
        raise RuntimeError(f"conv_gather: out {tuple(out.shape)} != {oshape}")
    for t, nm in ((mask, "mask"), (residual, "residual")):
        if t is not None and tuple(t.shape) != oshape:
            raise RuntimeError(f"conv_gather: {nm} {tuple(t.shape)} != {oshape}")
    if bias is not None and bias.numel() != d.Cout:
        raise RuntimeError("conv_gather: bias size")
    L.require_gpu_tensor(x, "x")
    wp, wn = _ws(lib.fcd_conv_workspace_bytes(C.byref(d)), x.device)
    timed = KERNEL_TIMER is not None and KERNEL_TIMER.match(d)
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.fcd_conv_gather(C.byref(d), ptr(x), ptr(w_packed), ptr(bias), ptr(mask), ptr(residual), ptr(out),
                              wp, wn, stream()), "conv_gather")
    if timed:
        e1.record()
        KERNEL_TIMER.pairs.append((e0, e1))
        KERNEL_TIMER.note(timed, d)
    return out


def conv_gather_stats(d: ConvDesc, x: torch.Tensor, w_packed: torch.Tensor, want_var: bool = False, apply=None):
    """Convolution + InstanceNorm statistics of its output: ``(z, mean, rstd, var_unbiased)``.

    ``apply=(gamma, beta, act_slope, residual, amax)``: also the norm's output ``y = act(IN(z)) (+ residual)`` and, with
    ``amax`` (zeroed AMAX_SLOTS-element tensor), ``max |y|``; returns ``(z, mean, rstd, var_unbiased, y)``."""
    lib = L.load()
    exp_in = (d.N, d.IH, d.IW, d.Cin)
    if tuple(x.shape) != exp_in:
        raise RuntimeError(f"conv_gather_stats: input {tuple(x.shape)} != descriptor {exp_in}")
    _check_packed(d, w_packed, d.KH * d.KW * d.Cin)
    L.require_gpu_tensor(x, "x")
    out = torch.empty((d.N, d.OH, d.OW, d.CoutP), dtype=torch.float32, device=x.device)
    mean = torch.empty((d.N, d.CoutP), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    varu = torch.empty_like(mean) if want_var else None
    wp, wn = _ws(lib.fcd_conv_gather_stats_workspace_bytes(C.byref(d)), x.device)
    timed = KERNEL_TIMER is not None and KERNEL_TIMER.match(d)
    if apply is not None:
        gamma, beta, act_slope, residual, amax = apply
        if gamma.numel() != d.CoutP or beta.numel() != d.CoutP:
            raise RuntimeError(f"conv_gather_stats: gamma / beta must have {d.CoutP} elements")
        if residual is not None and tuple(residual.shape) != tuple(out.shape):
            raise RuntimeError(f"conv_gather_stats: residual {tuple(residual.shape)} != output {tuple(out.shape)}")
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if apply is None or timed:
        # (a timed launch keeps the norm's output out of the bracket: the two calls, same bits)
        check(lib.fcd_conv_gather_stats(C.byref(d), ptr(x), ptr(w_packed), ptr(out), ptr(mean), ptr(rstd), ptr(varu), IN_EPS,
                                        wp, wn, stream()), "conv_gather_stats")
    else:
        y = torch.empty_like(out)
        check(lib.fcd_conv_gather_stats_apply(C.byref(d), ptr(x), ptr(w_packed), ptr(out), ptr(mean), ptr(rstd), ptr(varu),
                                              IN_EPS, ptr(gamma), ptr(beta), ptr(residual), ptr(y), act_slope,
                                              _amax_ptr(amax), wp, wn, stream()), "conv_gather_stats_apply")
    if timed:       # includes the (few microseconds) finalize kernel that follows the convolution
        e1.record()
        KERNEL_TIMER.pairs.append((e0, e1))
        KERNEL_TIMER.note(timed, d)
        if apply is not None:
            y = instnorm_apply(out, mean, rstd, gamma, beta, act_slope, residual=residual, amax=amax)
    if apply is not None:
        return out, mean, rstd, varu, y
    return out, mean, rstd, varu


def conv_wgrad(d: WgradDesc, gsmall: torch.Tensor, xbig: torch.Tensor, dw: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    if tuple(gsmall.shape) != (d.N, d.GH, d.GW, d.OP):
        raise RuntimeError(f"conv_wgrad: gsmall {tuple(gsmall.shape)} != {(d.N, d.GH, d.GW, d.OP)}")
    if tuple(xbig.shape) != (d.N, d.BH, d.BW, d.IP):
        raise RuntimeError(f"conv_wgrad: xbig {tuple(xbig.shape)} != {(d.N, d.BH, d.BW, d.IP)}")
    if dw is None:
        if d.accumulate:
            raise RuntimeError("conv_wgrad: accumulate needs dw")
        dw = torch.empty((d.O, d.I, d.KH, d.KW), dtype=torch.float32, device=xbig.device)
    elif dw.numel() != d.O * d.I * d.KH * d.KW:
        raise RuntimeError("conv_wgrad: dw size")
    L.require_gpu_tensor(gsmall, "gsmall")
    L.require_gpu_tensor(xbig, "xbig")
    L.require_gpu_tensor(dw, "dw")
    wp, wn = _ws(lib.fcd_wgrad_workspace_bytes(C.byref(d)), xbig.device)
    check(lib.fcd_conv_wgrad(C.byref(d), ptr(gsmall), ptr(xbig), ptr(dw), wp, wn, stream()), "conv_wgrad")
    return dw


def conv_wgrad2_supported(d: WgradDesc) -> bool:
    """True where ``conv_wgrad2`` has a kernel for the shape and precision of ``d``."""
    return bool(L.load().fcd_conv_wgrad2_supported(C.byref(d)))


def conv_wgrad2(d: WgradDesc, gsmall: torch.Tensor, xbig: torch.Tensor, gsmall2: torch.Tensor, xbig2: torch.Tensor,
                dw: Optional[torch.Tensor] = None, amax2=None) -> torch.Tensor:
    """``conv_wgrad`` over two operand pairs of one geometry (``d.N`` and ``gsmall2.shape[0]`` images) in one pass over
    ``dw``: ``dw (+)= gsmall^T xbig + gsmall2^T xbig2``.  ``amax2``: (max |gsmall2|, max |xbig2|) slot arrays, as
    ``set_wgrad_amax`` attaches the first pair's to ``d`` (the scaled-fp16 forms need all four)."""
    lib = L.load()
    n2 = gsmall2.shape[0]
    if tuple(gsmall.shape) != (d.N, d.GH, d.GW, d.OP) or tuple(xbig.shape) != (d.N, d.BH, d.BW, d.IP):
        raise RuntimeError("conv_wgrad2: first operand pair does not match the descriptor")
    if tuple(gsmall2.shape) != (n2, d.GH, d.GW, d.OP) or tuple(xbig2.shape) != (n2, d.BH, d.BW, d.IP):
        raise RuntimeError("conv_wgrad2: second operand pair does not match the descriptor's geometry")
    if dw is None:
        if d.accumulate:
            raise RuntimeError("conv_wgrad2: accumulate needs dw")
        dw = torch.empty((d.O, d.I, d.KH, d.KW), dtype=torch.float32, device=xbig.device)
    elif dw.numel() != d.O * d.I * d.KH * d.KW:
        raise RuntimeError("conv_wgrad2: dw size")
    for t, nm in ((gsmall, "gsmall"), (xbig, "xbig"), (gsmall2, "gsmall2"), (xbig2, "xbig2"), (dw, "dw")):
        L.require_gpu_tensor(t, nm)
    n1 = d.N
    d.N = n1 + n2
    need = lib.fcd_wgrad_workspace_bytes(C.byref(d))
    d.N = n1
    wp, wn = _ws(need, xbig.device)
    ga2, xa2 = (None, None) if amax2 is None else amax2
    check(lib.fcd_conv_wgrad2(C.byref(d), ptr(gsmall), ptr(xbig), ptr(gsmall2), ptr(xbig2), n2,
                              _amax_ptr(ga2), _amax_ptr(xa2), ptr(dw), wp, wn, stream()), "conv_wgrad2")
    return dw


def pack_weight_fwd(w: torch.Tensor, g: ConvGeom) -> torch.Tensor:
    L.require_gpu_tensor(w, "weight")
    out = torch.empty((g.O, g.k, g.k, g.IP), dtype=torch.float32, device=w.device)
    check(L.load().fcd_pack_weight_fwd(ptr(w), ptr(out), g.O, g.I, g.IP, g.k, g.k, stream()), "pack_weight_fwd")
    return out


def wants_split_weights(d: ConvDesc) -> bool:
    """True where a FCD_PREC_BF16X3 convolution takes three-plane bf16 weights (``*_x3`` packs)."""
    return bool(L.load().fcd_conv_wants_split_weights(C.byref(d)))


def pack_weight_fwd_h2(w: torch.Tensor, g: ConvGeom, w_amax: torch.Tensor) -> torch.Tensor:
    L.require_gpu_tensor(w, "weight")
    out = torch.empty((2, g.O, g.k, g.k, g.IP), dtype=torch.float16, device=w.device)
    check(L.load().fcd_pack_weight_fwd_h2(ptr(w), ptr(out), ptr(w_amax), g.O, g.I, g.IP, g.k, g.k, stream()), "pack_weight_fwd_h2")
    return out


def pack_weight_adj_h2(w: torch.Tensor, g: ConvGeom, w_amax: torch.Tensor) -> torch.Tensor:
    L.require_gpu_tensor(w, "weight")
    t = g.adj_taps()
    out = torch.empty((g.adj_phases(), 2, g.I, t, t, g.OP), dtype=torch.float16, device=w.device)
    check(L.load().fcd_pack_weight_adj_h2(ptr(w), ptr(out), _amax_ptr(w_amax), g.O, g.OP, g.I, g.k, g.k, g.s, stream()),
          "pack_weight_adj_h2")
    return out


PACK_BATCH_MAX = 16   # FCD_PACK_BATCH_MAX


def pack_weight_h2_batch(kind: str, ws, g: ConvGeom, w_amaxes):
    """``pack_weight_fwd_h2`` (kind "fwd") / ``pack_weight_adj_h2`` ("adj") of several weights of one geometry in one
    launch per PACK_BATCH_MAX weights; returns the list of packed tensors."""
    import ctypes
    lib = L.load()
    outs = []
    for w in ws:
        L.require_gpu_tensor(w, "weight")
        if tuple(w.shape) != tuple(ws[0].shape):
            raise RuntimeError("pack_weight_h2_batch: weights of different shapes")
        if kind == "fwd":
            outs.append(torch.empty((2, g.O, g.k, g.k, g.IP), dtype=torch.float16, device=w.device))
        else:
            t = g.adj_taps()
            outs.append(torch.empty((g.adj_phases(), 2, g.I, t, t, g.OP), dtype=torch.float16, device=w.device))
    for i in range(0, len(ws), PACK_BATCH_MAX):
        n = len(ws[i:i + PACK_BATCH_MAX])
        arr = ctypes.c_void_p * n
        wp = arr(*[ptr(w) for w in ws[i:i + n]])
        op = arr(*[ptr(o) for o in outs[i:i + n]])
        ap = arr(*[_amax_ptr(a) for a in w_amaxes[i:i + n]])
        if kind == "fwd":
            check(lib.fcd_pack_weight_fwd_h2_batch(wp, op, ap, n, g.O, g.I, g.IP, g.k, g.k, stream()), "pack_weight_fwd_h2_batch")
        else:
            check(lib.fcd_pack_weight_adj_h2_batch(wp, op, ap, n, g.O, g.OP, g.I, g.k, g.k, g.s, stream()),
                  "pack_weight_adj_h2_batch")
    return outs


def pack_weight_fwd_x3(w: torch.Tensor, g: ConvGeom) -> torch.Tensor:
    L.require_gpu_tensor(w, "weight")
    out = torch.empty((3, g.O, g.k, g.k, g.IP), dtype=torch.bfloat16, device=w.device)
    check(L.load().fcd_pack_weight_fwd_x3(ptr(w), ptr(out), g.O, g.I, g.IP, g.k, g.k, stream()), "pack_weight_fwd_x3")
    return out


def pack_weight_adj_x3(w: torch.Tensor, g: ConvGeom) -> torch.Tensor:
    L.require_gpu_tensor(w, "weight")
    t = g.adj_taps()
    out = torch.empty((g.adj_phases(), 3, g.I, t, t, g.OP), dtype=torch.bfloat16, device=w.device)
    check(L.load().fcd_pack_weight_adj_x3(ptr(w), ptr(out), g.O, g.OP, g.I, g.k, g.k, g.s, stream()), "pack_weight_adj_x3")
    return out


def pack_weight_adj(w: torch.Tensor, g: ConvGeom) -> torch.Tensor:
    L.require_gpu_tensor(w, "weight")
    t = g.adj_taps()
    out = torch.empty((g.adj_phases(), g.I, t, t, g.OP), dtype=torch.float32, device=w.device)
    check(L.load().fcd_pack_weight_adj(ptr(w), ptr(out), g.O, g.OP, g.I, g.k, g.k, g.s, stream()), "pack_weight_adj")
    return out


# --------------------------------------------------------------------------- #
# Discriminator heads
# --------------------------------------------------------------------------- #
def head_fwd(x: torch.Tensor, w_packed: torch.Tensor, O: int, k: int, pad: int) -> torch.Tensor:
    n, ih, iw, ip = x.shape
    oh, ow = ih + 2 * pad - k + 1, iw + 2 * pad - k + 1
    out = torch.empty((n, O, oh, ow), dtype=torch.float32, device=x.device)
    check(L.load().fcd_head_fwd(ptr(x), ptr(w_packed), ptr(out), n, ih, iw, ip, O, k, k, pad, stream()), "head_fwd")
    return out


def head_dgrad(gout: torch.Tensor, w_packed: torch.Tensor, din: torch.Tensor, mask, mask_slope: float,
               accumulate: bool, k: int, pad: int) -> torch.Tensor:
    n, ih, iw, ip = din.shape
    O = gout.shape[1]
    check(L.load().fcd_head_dgrad(ptr(gout), ptr(w_packed), ptr(din), ptr(mask), mask_slope, int(accumulate), n, ih, iw,
                                  ip, O, k, k, pad, stream()), "head_dgrad")
    return din


def head_wgrad(gout: torch.Tensor, x: torch.Tensor, I: int, k: int, pad: int, dw: Optional[torch.Tensor] = None,
               accumulate: bool = False) -> torch.Tensor:
    n, ih, iw, ip = x.shape
    O = gout.shape[1]
    if dw is None:
        dw = torch.empty((O, I, k, k), dtype=torch.float32, device=x.device)
    check(L.load().fcd_head_wgrad(ptr(gout), ptr(x), ptr(dw), int(accumulate), n, ih, iw, I, ip, O, k, k, pad, stream()),
          "head_wgrad")
    return dw


# --------------------------------------------------------------------------- #
# instance norm
# --------------------------------------------------------------------------- #
def instnorm_stats(z: torch.Tensor, want_var: bool = False):
    n, h, w, c = z.shape
    lib = L.load()
    mean = torch.empty((n, c), dtype=torch.float32, device=z.device)
    rstd = torch.empty_like(mean)
    varu = torch.empty_like(mean) if want_var else None
    wp, wn = _ws(lib.fcd_instnorm_workspace_bytes(n, h * w, c), z.device)
    check(lib.fcd_instnorm_stats(ptr(z), ptr(mean), ptr(rstd), ptr(varu), n, h * w, c, IN_EPS, wp, wn, stream()),
          "instnorm_stats")
    return mean, rstd, varu


def instnorm_stats_from_running(rm: torch.Tensor, rv: torch.Tensor, n: int):
    c = rm.numel()
    mean = torch.empty((n, c), dtype=torch.float32, device=rm.device)
    rstd = torch.empty_like(mean)
    check(L.load().fcd_instnorm_stats_from_running(ptr(rm), ptr(rv), ptr(mean), ptr(rstd), n, c, IN_EPS, stream()),
          "stats_from_running")
    return mean, rstd


def instnorm_update_running(rm, rv, mean, varu, groups: int = 1):
    n, c = mean.shape
    check(L.load().fcd_instnorm_update_running(ptr(rm), ptr(rv), ptr(mean), ptr(varu), n, c, int(groups), IN_MOMENTUM,
                                               stream()), "update_running")


RUNNING_BATCH_MAX = 16


def instnorm_update_running_batch(items, groups: int = 1) -> None:
    """One launch for the momentum updates of several tracked norms of one generator call: ``items`` is a
    list of (running_mean, running_var, mean, var_unbiased) with equal (N, C)."""
    import ctypes
    for i in range(0, len(items), RUNNING_BATCH_MAX):
        part = items[i:i + RUNNING_BATCH_MAX]
        n, c = part[0][2].shape
        if any(it[2].shape != (n, c) for it in part):
            raise RuntimeError("instnorm_update_running_batch: norms of different (N, C) in one batch")
        arr = ctypes.c_void_p * len(part)
        cols = [arr(*[ptr(it[k]) for it in part]) for k in range(4)]
        check(L.load().fcd_instnorm_update_running_batch(cols[0], cols[1], cols[2], cols[3], len(part), n, c,
                                                         int(groups), IN_MOMENTUM, stream()), "update_running_batch")


AMAX_SLOTS = 1024    # FCD_AMAX_SLOTS: an "amax" is 1024 partial maxima, one per block of the kernel that produced it


def absmax(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``max |x|`` as a device "scalar" of AMAX_SLOTS partial maxima (one launch, no host sync); ``out`` (zeroed, or
    holding earlier maxima) is updated in place."""
    L.require_gpu_tensor(x, "x")
    if out is None:
        out = torch.zeros(AMAX_SLOTS, dtype=torch.float32, device=x.device)
    xc = x.contiguous()
    if xc.numel() % 4 or xc.data_ptr() % 16:
        raise ValueError("absmax: the tensor's length and address must be multiples of 4 floats")
    _amax_ptr(out)
    check(L.load().fcd_absmax(ptr(xc), xc.numel(), ptr(out), stream()), "absmax")
    return out


ABSMAX_BATCH_MAX = 32   # FCD_ABSMAX_BATCH_MAX


def absmax_batch(tensors, out: torch.Tensor) -> torch.Tensor:
    """Row ``i`` of ``out`` (``len(tensors)`` x AMAX_SLOTS, float32, contiguous) := the partial maxima of ``|tensors[i]|``;
    one launch per ABSMAX_BATCH_MAX tensors, rows are overwritten (no zeroing needed)."""
    import ctypes
    if out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous() or tuple(out.shape) != (len(tensors), AMAX_SLOTS):
        raise ValueError(f"absmax_batch: out must be a contiguous float32 device tensor of shape ({len(tensors)}, {AMAX_SLOTS})")
    for t in tensors:
        L.require_gpu_tensor(t, "tensor")
        if t.numel() % 4 or t.data_ptr() % 16:
            raise RuntimeError("absmax_batch: a tensor's length / address is not a multiple of 4 floats")
    for i in range(0, len(tensors), ABSMAX_BATCH_MAX):
        part = tensors[i:i + ABSMAX_BATCH_MAX]
        xs = (ctypes.c_void_p * len(part))(*[t.data_ptr() for t in part])
        ns = (ctypes.c_size_t * len(part))(*[t.numel() for t in part])
        check(L.load().fcd_absmax_batch(xs, ns, len(part), out[i].data_ptr(), stream()), "absmax_batch")
    return out


def instnorm_apply(z, mean, rstd, gamma, beta, act_slope: float, residual=None, out=None, amax=None):
    """``amax`` (a zeroed AMAX_SLOTS-element device tensor): also receives ``max |out|`` (PREC_F16X2 operand scaling)."""
    n, h, w, c = z.shape
    if out is None:
        out = torch.empty_like(z)
    check(L.load().fcd_instnorm_apply_amax(ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(residual), ptr(out), n,
                                           h * w, c, act_slope, _amax_ptr(amax), stream()), "instnorm_apply")
    return out


def instnorm_bwd(gy, y, z, mean, rstd, gamma, act_slope: float, use_running: bool, want_affine: bool = True,
                 dgamma=None, dbeta=None, accumulate: bool = False, beta=None, dz_amax=None):
    """``dgamma`` / ``dbeta`` given: written (``accumulate``: added to) in place.  ``beta`` given:
    the activation mask is recomputed from ``z`` and ``y`` may be None.  ``dz_amax`` (zeroed AMAX_SLOTS-element device
    tensor): also receives ``max |dz|``."""
    n, h, w, c = z.shape
    lib = L.load()
    dz = torch.empty_like(z)
    if want_affine and dgamma is None and dbeta is None:
        if accumulate:
            raise RuntimeError("instnorm_bwd: accumulate needs dgamma / dbeta")
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
    wp, wn = _ws(lib.fcd_instnorm_workspace_bytes(n, h * w, c), z.device)
    check(lib.fcd_instnorm_bwd_amax(ptr(gy), ptr(y), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dz), ptr(dgamma),
                                    ptr(dbeta), int(accumulate), n, h * w, c, act_slope, int(use_running), _amax_ptr(dz_amax),
                                    wp, wn, stream()), "instnorm_bwd")
    return dz, dgamma, dbeta


AFFINE_BATCH_MAX = 32   # FCD_AFFINE_BATCH_MAX


def instnorm_bwd_deferred(gy, z, mean, rstd, gamma, act_slope: float, use_running: bool, beta, dz_amax=None):
    """The backward with the affine gradients left for :func:`instnorm_affine_batch`: returns ``(dz, s12)`` with
    ``s12[n][c] = (sum g1, sum g1 * xhat)``."""
    n, h, w, c = z.shape
    lib = L.load()
    dz = torch.empty_like(z)
    s12 = torch.empty((n, c, 2), dtype=torch.float32, device=z.device)
    wp, wn = _ws(lib.fcd_instnorm_workspace_bytes(n, h * w, c), z.device)
    check(lib.fcd_instnorm_bwd_s12(ptr(gy), None, ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dz), ptr(s12), n,
                                   h * w, c, act_slope, int(use_running), _amax_ptr(dz_amax), wp, wn, stream()),
          "instnorm_bwd_s12")
    return dz, s12


def instnorm_affine_batch(items, accumulate: bool) -> None:
    """``dgamma (+)= sum_n s12[n, :, 1]``, ``dbeta (+)= sum_n s12[n, :, 0]`` for a list of ``(s12, dgamma, dbeta)``:
    one launch per group of equal (N, C) (at most AFFINE_BATCH_MAX norms each)."""
    import ctypes
    groups = {}
    for it in items:
        groups.setdefault(tuple(it[0].shape), []).append(it)
    for (n, c, _), its in groups.items():
        for i in range(0, len(its), AFFINE_BATCH_MAX):
            part = its[i:i + AFFINE_BATCH_MAX]
            for s12, dg, db in part:
                for t in (dg, db):
                    if t is not None and (t.numel() != c or t.dtype != torch.float32 or not t.is_contiguous()):
                        raise RuntimeError("instnorm_affine_batch: a gradient is not a contiguous float32 tensor of C elements")
            arr = ctypes.c_void_p * len(part)
            cols = [arr(*[ptr(it[k]) for it in part]) for k in range(3)]
            check(L.load().fcd_instnorm_affine_batch(cols[0], cols[1], cols[2], len(part), n, c, int(accumulate), stream()),
                  "instnorm_affine_batch")


# --------------------------------------------------------------------------- #
# layout / Generator prologue + epilogue
# --------------------------------------------------------------------------- #
def g_prologue(x: torch.Tensor, c: torch.Tensor, cp: int) -> torch.Tensor:
    n, ch, h, w = x.shape
    out = torch.empty((n, h, w, cp), dtype=torch.float32, device=x.device)
    check(L.load().fcd_g_prologue(ptr(x), ptr(c), ptr(out), n, ch, c.shape[1], cp, h, w, stream()), "g_prologue")
    return out


def nchw_to_nhwc(x: torch.Tensor, cp: int) -> torch.Tensor:
    n, ch, h, w = x.shape
    out = torch.empty((n, h, w, cp), dtype=torch.float32, device=x.device)
    check(L.load().fcd_nchw_to_nhwc(ptr(x), ptr(out), n, ch, cp, h, w, stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, c: int, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    n, h, w, cp = x.shape
    if out is None:
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(L.load().fcd_nhwc_to_nchw(ptr(x), ptr(out), int(accumulate), n, c, cp, h, w, stream()), "nhwc_to_nchw")
    return out


def g_epilogue_fwd(delta: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    n, ch, h, w = x.shape
    out = torch.empty_like(x)
    check(L.load().fcd_g_epilogue_fwd(ptr(delta), ptr(x), ptr(out), n, ch, delta.shape[3], h, w, stream()), "g_epilogue_fwd")
    return out


def g_epilogue_bwd(gout, x, x_fake, cp: int, gx: Optional[torch.Tensor]):
    n, ch, h, w = x.shape
    gdelta = torch.empty((n, h, w, cp), dtype=torch.float32, device=x.device)
    check(L.load().fcd_g_epilogue_bwd(ptr(gout), ptr(x), ptr(x_fake), ptr(gdelta), ptr(gx), 0, n, ch, cp, h, w, stream()),
          "g_epilogue_bwd")
    return gdelta


# --------------------------------------------------------------------------- #
# elementwise / reductions
# --------------------------------------------------------------------------- #
def act_fwd(z: torch.Tensor, act: float) -> torch.Tensor:
    a = torch.empty_like(z)
    check(L.load().fcd_act_fwd(ptr(z), ptr(a), z.numel(), act, stream()), "act_fwd")
    return a


def act_bwd(g: torch.Tensor, z: torch.Tensor, act: float, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """out (+)= g * act'(z)."""
    if out is None:
        out = torch.empty_like(z)
    check(L.load().fcd_act_bwd(ptr(g), ptr(z), ptr(out), z.numel(), act, int(accumulate), stream()), "act_bwd")
    return out


def act_bwd2(ubar: torch.Tensor, ga: torch.Tensor, z: torch.Tensor, act: float):
    """(ubar * act'(z), ubar * ga * act''(z))."""
    u, q = torch.empty_like(z), torch.empty_like(z)
    check(L.load().fcd_act_bwd2(ptr(ubar), ptr(ga), ptr(z), ptr(u), ptr(q), z.numel(), act, stream()), "act_bwd2")
    return u, q


def colsum(x2d_rows: int, x: torch.Tensor, c: int, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    lib = L.load()
    if out is None:
        out = torch.empty((c,), dtype=torch.float32, device=x.device)
    wp, wn = _ws(lib.fcd_colsum_workspace_bytes(x2d_rows, c), x.device)
    check(lib.fcd_colsum(ptr(x), ptr(out), int(accumulate), x2d_rows, c, wp, wn, stream()), "colsum")
    return out


def _red_ws(device):
    return _ws(L.load().fcd_reduce_workspace_bytes(), device)


def sum_scaled(x: torch.Tensor, scale: float) -> torch.Tensor:
    out = torch.empty((), dtype=torch.float32, device=x.device)
    wp, wn = _red_ws(x.device)
    check(L.load().fcd_sum_scaled(ptr(x), ptr(out), x.numel(), scale, wp, wn, stream()), "sum_scaled")
    return out


def fill_scaled(shape, scale: float, gscale: torch.Tensor) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=gscale.device)
    check(L.load().fcd_fill_scaled(ptr(out), out.numel(), scale, ptr(gscale), stream()), "fill_scaled")
    return out


def l1_mean(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty((), dtype=torch.float32, device=a.device)
    wp, wn = _red_ws(a.device)
    check(L.load().fcd_l1_mean(ptr(a), ptr(b), ptr(out), a.numel(), wp, wn, stream()), "l1_mean")
    return out


def l1_mean_bwd(a, b, gscale, want_a: bool, want_b: bool):
    ga = torch.empty_like(a) if want_a else None
    gb = torch.empty_like(b) if want_b else None
    check(L.load().fcd_l1_mean_bwd(ptr(a), ptr(b), ptr(gscale), ptr(ga), ptr(gb), 0, a.numel(), stream()), "l1_mean_bwd")
    return ga, gb


def bce_logits_mean(logit, target) -> torch.Tensor:
    out = torch.empty((), dtype=torch.float32, device=logit.device)
    wp, wn = _red_ws(logit.device)
    check(L.load().fcd_bce_logits_mean(ptr(logit), ptr(target), ptr(out), logit.numel(), wp, wn, stream()), "bce_logits_mean")
    return out


def bce_logits_mean_bwd(logit, target, gscale):
    g = torch.empty_like(logit)
    check(L.load().fcd_bce_logits_mean_bwd(ptr(logit), ptr(target), ptr(gscale), ptr(g), logit.numel(), stream()),
          "bce_logits_mean_bwd")
    return g


def lerp_rows(a, b, alpha):
    out = torch.empty_like(a)
    n = a.shape[0]
    check(L.load().fcd_lerp_rows(ptr(a), ptr(b), ptr(alpha), ptr(out), n, a.numel() // n, stream()), "lerp_rows")
    return out


def gp_fwd(g: torch.Tensor):
    n = g.shape[0]
    norm = torch.empty((n,), dtype=torch.float32, device=g.device)
    out = torch.empty((), dtype=torch.float32, device=g.device)
    check(L.load().fcd_gp_fwd(ptr(g), ptr(norm), ptr(out), n, g.numel() // n, stream()), "gp_fwd")
    return out, norm


def gp_bwd(g, norm, gscale):
    gg = torch.empty_like(g)
    n = g.shape[0]
    check(L.load().fcd_gp_bwd(ptr(g), ptr(norm), ptr(gscale), ptr(gg), n, g.numel() // n, stream()), "gp_bwd")
    return gg


def adam_step(p, g, m, v, lr: float, beta1: float, beta2: float, eps: float, step: int) -> None:
    check(L.load().fcd_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, stream()),
          "adam_step")


def diffmap_threshold(x_fake, x, thr: Optional[float], want_diff: bool = True):
    n, c, h, w = x.shape
    diff = torch.empty((n, h, w), dtype=torch.float32, device=x.device) if want_diff else None
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if thr is not None else None
    check(L.load().fcd_diffmap_threshold(ptr(x_fake), ptr(x), ptr(diff), ptr(mask), 0.0 if thr is None else float(thr),
                                         n, c, h, w, stream()), "diffmap_threshold")
    return diff, mask


# --------------------------------------------------------------------------- #
# scene path / threshold search on the device
# --------------------------------------------------------------------------- #
def scene_tiles(scene: torch.Tensor, origins: torch.Tensor, patch: int):
    """scene (H, W, C) uint16 and origins (nb, 2) int32 on the device -> x (nb, C, patch, patch)
    fp32 in [-1, 1] and a device int that is 1 if any sample of the batch is non-zero."""
    if scene.dtype != torch.uint16 or origins.dtype != torch.int32:
        raise TypeError("scene_tiles: scene must be uint16 and origins int32")
    L.require_gpu_tensor(scene, "scene", torch.uint16)
    L.require_gpu_tensor(origins, "origins", torch.int32)
    h, w, c = scene.shape
    nb = origins.shape[0]
    x = torch.empty((nb, c, patch, patch), dtype=torch.float32, device=scene.device)
    flag = torch.zeros((), dtype=torch.int32, device=scene.device)
    check(L.load().fcd_scene_tiles(ptr(scene), h, w, c, patch, ptr(origins), nb, ptr(x), ptr(flag), stream()), "scene_tiles")
    return x, flag


def scene_tile_valid(scene: torch.Tensor, origins: torch.Tensor, patch: int) -> torch.Tensor:
    """(nb,) uint8 on the device: 1 where tile b of the scene holds any non-zero sample."""
    L.require_gpu_tensor(scene, "scene", torch.uint16)
    L.require_gpu_tensor(origins, "origins", torch.int32)
    h, w, c = scene.shape
    nb = origins.shape[0]
    valid = torch.empty((nb,), dtype=torch.uint8, device=scene.device)
    check(L.load().fcd_scene_tile_valid(ptr(scene), h, w, c, patch, ptr(origins), nb, ptr(valid), stream()), "scene_tile_valid")
    return valid


def scene_tif_bytes(raster: torch.Tensor, prediction: Optional[torch.Tensor], valid: Optional[torch.Tensor] = None,
                    scene: Optional[torch.Tensor] = None, want_diff: bool = True):
    """``write_tifs``' byte rasters: ((difference * 255) as uint8, mask in {0, 128, 255}); either may be None."""
    h, w = raster.shape
    L.require_gpu_tensor(raster, "raster", torch.float64)
    for t, nm in ((prediction, "prediction"), (valid, "valid")):
        if t is not None:
            L.require_gpu_tensor(t, nm, torch.uint8)
            if tuple(t.shape) != (h, w):
                raise RuntimeError(f"scene_tif_bytes: {nm} must be (H, W)")
    c = 0
    if scene is not None:
        L.require_gpu_tensor(scene, "scene", torch.uint16)
        if tuple(scene.shape[:2]) != (h, w):
            raise RuntimeError("scene_tif_bytes: scene must be (H, W, C)")
        c = scene.shape[2]
    diff_u8 = torch.empty((h, w), dtype=torch.uint8, device=raster.device) if want_diff else None
    mask_u8 = torch.empty((h, w), dtype=torch.uint8, device=raster.device) if prediction is not None else None
    check(L.load().fcd_scene_tif_bytes(ptr(raster), ptr(prediction), ptr(valid), ptr(scene), h, w, c, ptr(diff_u8),
                                       ptr(mask_u8), stream()), "scene_tif_bytes")
    return diff_u8, mask_u8


def pseudo_mask(diff: torch.Tensor, target: Optional[torch.Tensor], thr: float, out: Optional[torch.Tensor] = None,
                cm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``(diff > thr) + 1`` with invalid pixels (``target == 0``) set to 0 (``fcd_solver.py:762-772``);
    ``cm`` (2, 2) int64 on the device, when given, gets the confusion counts of the valid pixels added."""
    L.require_gpu_tensor(diff, "diff")
    if target is not None:
        L.require_gpu_tensor(target, "target", torch.uint8)
        if target.numel() != diff.numel():
            raise RuntimeError("pseudo_mask: target and diff differ in size")
    if out is None:
        out = torch.empty(diff.shape, dtype=torch.uint8, device=diff.device)
    else:
        L.require_gpu_tensor(out, "out", torch.uint8)
        if out.numel() != diff.numel():
            raise RuntimeError("pseudo_mask: out and diff differ in size")
    if cm is not None:
        L.require_gpu_tensor(cm, "cm", torch.int64)
        if cm.numel() != 4:
            raise RuntimeError("pseudo_mask: cm must be (2, 2) int64")
    check(L.load().fcd_pseudo_mask(ptr(diff), ptr(target), diff.numel(), float(thr), ptr(out), ptr(cm), stream()), "pseudo_mask")
    return out


def scene_stitch(delta: torch.Tensor, origins: torch.Tensor, raster: torch.Tensor) -> None:
    nb, patch, _ = delta.shape
    h, w = raster.shape
    if raster.dtype != torch.float64 or delta.dtype != torch.float32:
        raise TypeError("scene_stitch: delta fp32, raster float64")
    L.require_gpu_tensor(delta, "delta")
    L.require_gpu_tensor(raster, "raster", torch.float64)
    L.require_gpu_tensor(origins, "origins", torch.int32)
    check(L.load().fcd_scene_stitch(ptr(delta), ptr(origins), nb, patch, ptr(raster), h, w, stream()), "scene_stitch")


def scene_binarize(raster: torch.Tensor, scene: torch.Tensor, threshold: float) -> torch.Tensor:
    h, w, c = scene.shape
    if tuple(raster.shape) != (h, w) or raster.dtype != torch.float64:
        raise RuntimeError("scene_binarize: raster must be (H, W) float64")
    L.require_gpu_tensor(raster, "raster", torch.float64)
    L.require_gpu_tensor(scene, "scene", torch.uint16)
    mask = torch.empty((h, w), dtype=torch.uint8, device=raster.device)
    check(L.load().fcd_scene_binarize(ptr(raster), ptr(scene), h, w, c, float(threshold), ptr(mask), stream()),
          "scene_binarize")
    return mask


def valid_minmax(diff: torch.Tensor, target: torch.Tensor, minmax: torch.Tensor) -> None:
    """Fold min / max of ``diff`` where ``target > 0`` into ``minmax`` (2 fp32, start at [inf, 0])."""
    if diff.numel() != target.numel() or target.dtype != torch.uint8 or diff.dtype != torch.float32:
        raise RuntimeError("valid_minmax: diff fp32 and target uint8 of the same size")
    L.require_gpu_tensor(diff, "diff")
    L.require_gpu_tensor(target, "target", torch.uint8)
    L.require_gpu_tensor(minmax, "minmax")
    check(L.load().fcd_valid_minmax(ptr(diff), ptr(target), diff.numel(), ptr(minmax), stream()), "valid_minmax")


def threshold_hist(diff: torch.Tensor, target: torch.Tensor, thresholds: torch.Tensor, hist: torch.Tensor) -> None:
    """Add this batch to ``hist`` (2, T+1) int64: see ``fcd_threshold_hist`` in include/fcd_hip.h."""
    t = thresholds.numel()
    if hist.dtype != torch.int64 or tuple(hist.shape) != (2, t + 1) or thresholds.dtype != torch.float32:
        raise RuntimeError("threshold_hist: hist must be (2, T+1) int64 and thresholds fp32")
    if diff.numel() != target.numel() or target.dtype != torch.uint8 or diff.dtype != torch.float32:
        raise RuntimeError("threshold_hist: diff fp32 and target uint8 of the same size")
    L.require_gpu_tensor(diff, "diff")
    L.require_gpu_tensor(target, "target", torch.uint8)
    L.require_gpu_tensor(hist, "hist", torch.int64)
    check(L.load().fcd_threshold_hist(ptr(diff), ptr(target), diff.numel(), ptr(thresholds), t, ptr(hist), stream()),
          "threshold_hist")


# --------------------------------------------------------------------------- #
# input pipeline on the device
# --------------------------------------------------------------------------- #
def patch_batch(images: torch.Tensor, index: torch.Tensor, aug: Optional[torch.Tensor] = None,
                masks: Optional[torch.Tensor] = None):
    """images (N, H, W, C) uint16, index (B,) int64, aug (B, 3) int32 {flags, dy, dx} or None and
    optional masks (N, H, W) uint8, all on the device -> x (B, C, H, W) fp32 in [-1, 1] and the
    gathered masks (B, H, W) uint8 (or None)."""
    L.require_gpu_tensor(images, "images", torch.uint16)
    L.require_gpu_tensor(index, "index", torch.int64)
    if images.dim() != 4 or index.dim() != 1:
        raise ValueError("patch_batch: images must be (N, H, W, C) and index (B,)")
    n, h, w, c = images.shape
    b = index.shape[0]
    if aug is not None:
        L.require_gpu_tensor(aug, "aug", torch.int32)
        if tuple(aug.shape) != (b, 3):
            raise ValueError("patch_batch: aug must be (B, 3)")
    mask_out = None
    if masks is not None:
        L.require_gpu_tensor(masks, "masks", torch.uint8)
        if tuple(masks.shape) != (n, h, w):
            raise ValueError("patch_batch: masks must be (N, H, W)")
        mask_out = torch.empty((b, h, w), dtype=torch.uint8, device=images.device)
    x = torch.empty((b, c, h, w), dtype=torch.float32, device=images.device)
    check(L.load().fcd_patch_batch(ptr(images), n, h, w, c, ptr(index), ptr(aug), b, ptr(x), ptr(masks), ptr(mask_out),
                                   stream()), "patch_batch")
    return x, mask_out
