"""StarGAN-style Generator of the Fixed-Point GAN on the HIP kernels.

Drop-in for the reference ``Generator`` (``models/fixed_point_gan.py:37-84``,
``ResidualBlock`` ``:22-34``): same constructor, same ``forward(x, c)``, same
state_dict keys and shapes (``main.0.weight`` ... ``main.{9+r}.main.{0,3}.weight``,
``main.{9+r}.main.{1,4}.{weight,bias,running_mean,running_var,num_batches_tracked}``),
so a reference checkpoint loads unchanged.

The whole network is ONE autograd node: forward runs the chain of HIP launches
(implicit-GEMM convs on MFMA, fused InstanceNorm+ReLU(+residual), fused
tanh/invalid-mask epilogue) and keeps the per-layer tensors; backward runs the
hand-written reverse chain (InstanceNorm backward, data gradients as adjoint
convolutions, weight gradients as split-K GEMMs).  Activations inside are NHWC.
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .arena import grad_sink
from .ops import ConvGeom


def _act_slope(activation: str) -> float:
    if activation == "relu":
        return 0.0
    if activation == "lrelu":
        return ops.LRELU_SLOPE
    # nn.SiLU / nn.Mish (fixed_point_gan.py:13-16): activation codes of the InstanceNorm kernels
    # (generator only; the discriminator rejects them, see discriminator.py)
    if activation == "silu":
        return ops.ACT_SILU
    if activation == "mish":
        return ops.ACT_MISH
    raise NotImplementedError(f"activation {activation!r} has no HIP kernel")


class WeightMaxima:
    """max |weight| of all convolutions of a network from ONE launch per set of parameter versions
    (``fcd_absmax_batch``) instead of a zero-fill and a launch per layer: 17 + 17 small launches per generator step.
    ``refresh()`` at the start of a forward pass; ``row()`` answers only for the versions it was refreshed with (a
    convolution whose weight changed since, or whose weight the batched kernel does not take, falls back to its own
    ``absmax``).  ``refresh()`` also re-packs, one launch per shape and direction (``fcd_pack_weight_*_h2_batch``), the
    two-fp16-plane operands the layers asked for under the previous versions (``ConvParams._h2_wanted``); a layer
    that asks for something else packs it itself, as before."""

    def __init__(self, convs):
        self.convs = list(convs)
        self.keys = None
        self.rows = {}          # conv index -> row of buf
        self.buf = None
        for i, c in enumerate(self.convs):
            # plain attributes (the pool is neither a submodule nor a buffer); the index survives a deepcopy
            object.__setattr__(c, "_amax_pool", self)
            object.__setattr__(c, "_amax_index", i)

    def refresh(self) -> None:
        keys = [c._key() for c in self.convs]
        if keys == self.keys:
            return
        dev = self.convs[0].weight.device
        if self.buf is None or self.buf.device != dev:
            self.buf = torch.empty((len(self.convs), ops.AMAX_SLOTS), dtype=torch.float32, device=dev)
        with torch.no_grad():
            ws = [(i, c.weight.detach().contiguous()) for i, c in enumerate(self.convs)]
            ws = [(i, w) for i, w in ws if w.numel() % 4 == 0 and w.data_ptr() % 16 == 0]
            self.rows = {i: r for r, (i, _) in enumerate(ws)}
            if ws:
                ops.absmax_batch([w for _, w in ws], self.buf[:len(ws)])
            # the two-fp16-plane operands the layers asked for last time ("fwdh" / "adjh"), one launch per shape
            groups = {}
            for i, w in ws:
                c = self.convs[i]
                for kind, geom in getattr(c, "_h2_wanted", {}).items():
                    hit = c._packed.get(kind)
                    if hit is None or hit[0] != keys[i]:
                        gk = (kind, tuple(w.shape), geom.O, geom.I, geom.k, geom.s, geom.p, geom.OP, geom.IP)
                        groups.setdefault(gk, []).append((i, w, geom))
            for (kind, *_), items in groups.items():
                rows = [self.buf[self.rows[i]] for i, _, _ in items]
                outs = ops.pack_weight_h2_batch(kind[:3], [w for _, w, _ in items], items[0][2], rows)
                for (i, _, _), t, wa in zip(items, outs, rows):
                    self.convs[i]._packed[kind] = (keys[i], (t, wa))
        self.keys = keys

    def row(self, conv, key):
        i = getattr(conv, "_amax_index", None)
        if i is None or i >= len(self.convs) or self.convs[i] is not conv or self.keys is None or self.keys[i] != key:
            return None
        r = self.rows.get(i)
        return None if r is None else self.buf[r]


class ConvParams(nn.Module):
    """Holds a conv / transposed-conv weight (and bias); never called directly."""

    def __init__(self, shape, fan_in: int, bias: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(shape[0])) if bias else None
        self.fan_in = fan_in
        self.reset_parameters()
        self._packed = {}

    def reset_parameters(self):
        # torch's default conv init: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
        bound = 1.0 / math.sqrt(self.fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def operand(self, kind: str, geom: ConvGeom, d) -> torch.Tensor:
        """The packed weights convolution ``d`` wants: fp32, three bf16 planes (BF16X3, 128x128 tiles), or two fp16
        planes of the scaled weights (PREC_F16X2; also sets ``d.w_amax``, as for the 7x7 halo kernels that split
        fp32 weights themselves)."""
        fmt = ops.weight_format(d)
        if fmt == 2:
            t, wa = self.packed(kind + "h", geom)
            d.w_amax = wa.data_ptr()
            return t
        if fmt == 3:
            d.w_amax = self.weight_amax().data_ptr()
        return self.packed(kind + "3" if fmt == 1 else kind, geom)

    def _key(self):
        w = self.weight
        return (w.data_ptr(), w._version, str(w.device))

    def weight_amax(self) -> torch.Tensor:
        """max |weight| (AMAX_SLOTS partial maxima on the device), once per parameter version."""
        key = self._key()
        pool = getattr(self, "_amax_pool", None)
        if pool is not None:
            row = pool.row(self, key)        # the network's weights in one launch per optimizer step
            if row is not None:
                return row
        hit = self._packed.get("amax")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, ops.absmax(self.weight.detach().contiguous()))
            self._packed["amax"] = hit
        return hit[1]

    def packed(self, kind: str, geom: ConvGeom) -> torch.Tensor:
        """Operand layout for the gather kernel, rebuilt when the parameter changed."""
        w = self.weight
        key = self._key()
        hit = self._packed.get(kind)
        if hit is not None and hit[0] == key:
            return hit[1]
        if kind.endswith("h"):
            # remembered for WeightMaxima.refresh: the next set of parameter versions is packed for all layers of this
            # shape in one launch
            self.__dict__.setdefault("_h2_wanted", {})[kind] = geom
        with torch.no_grad():
            wd = w.detach().contiguous()
            if kind.endswith("h"):
                wa = self.weight_amax()
                t = ((ops.pack_weight_fwd_h2 if kind == "fwdh" else ops.pack_weight_adj_h2)(wd, geom, wa), wa)
            else:
                t = {"fwd": ops.pack_weight_fwd, "adj": ops.pack_weight_adj, "fwd3": ops.pack_weight_fwd_x3,
                     "adj3": ops.pack_weight_adj_x3}[kind](wd, geom)
        self._packed[kind] = (key, t)
        return t

    def folded(self, geom: ConvGeom, d, norm, epoch: int = 0) -> tuple:
        """Inference only: the forward operand with a tracked InstanceNorm's eval-mode transform folded
        in.  With running statistics the norm is the per-channel affine map ``z * s + t``,
        ``s = gamma / sqrt(running_var + eps)``, ``t = beta - running_mean * s`` (what
        ``nn.InstanceNorm2d(track_running_stats=True)`` computes in eval mode, fixed_point_gan.py:28,31), so ``norm(conv(a, W)) == conv(a, W * s) + t``: the scaled weights
        are packed once per set of parameter / buffer versions (``epoch``: the generator's count of
        kernel-side running-statistics updates, which bypass tensor versions), ``t`` goes in as the bias, the activation
        and the residual add ride in the convolution's epilogue, and no norm kernel runs at all."""
        srcs = (self.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var)
        fmt = ops.weight_format(d)       # 0 fp32, 1 three bf16 planes, 2 two fp16 planes + max |w|, 3 fp32 + max |w|
        key = tuple((t.data_ptr(), t._version) for t in srcs) + (fmt, str(self.weight.device), epoch)
        hit = self._packed.get("folded")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                scale = norm.weight.detach() / torch.sqrt(norm.running_var + ops.IN_EPS)
                shift = (norm.bias.detach() - norm.running_mean * scale).contiguous()
                wd = (self.weight.detach() * scale.view(-1, 1, 1, 1)).contiguous()
                wa = ops.absmax(wd) if fmt >= 2 else None
                if fmt == 2:
                    wp = ops.pack_weight_fwd_h2(wd, geom, wa)
                else:
                    wp = (ops.pack_weight_fwd_x3 if fmt == 1 else ops.pack_weight_fwd)(wd, geom)
            hit = (key, (wp, shift, wa))
            self._packed["folded"] = hit
        wp, shift, wa = hit[1]
        if wa is not None:
            d.w_amax = wa.data_ptr()
        return wp, shift


class NormParams(nn.Module):
    """InstanceNorm2d(affine=True[, track_running_stats=True]) parameters and buffers."""

    def __init__(self, ch: int, track: bool):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.track = track
        if track:
            self.register_buffer("running_mean", torch.zeros(ch))
            self.register_buffer("running_var", torch.ones(ch))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Slot(nn.Module):
    """Parameter-free placeholder that keeps the reference's Sequential indices."""


class ResidualBlock(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.main = nn.Sequential(
            ConvParams((dim_out, dim_in, 3, 3), dim_in * 9),
            NormParams(dim_out, True),
            _Slot(),
            ConvParams((dim_out, dim_out, 3, 3), dim_out * 9),
            NormParams(dim_out, True))


class _Layer:
    """One conv(+norm) stage of the plan: geometry + parameter holders."""

    def __init__(self, kind, geom, conv, norm, act_slope, res_from=None):
        self.kind = kind            # "conv" (eq-conv forward) or "convT" (its adjoint)
        self.geom = geom
        self.conv = conv
        self.norm = norm
        self.act_slope = act_slope  # activation after the norm (1.0 = none)
        self.res_from = res_from    # index into the saved activations for the residual add


class Generator(nn.Module):
    """Generator network (HIP).  Reference: ``models/fixed_point_gan.py:37-84``."""

    def __init__(self, conv_dim=64, c_dim=5, repeat_num=6, num_channels=3, activation="relu"):
        super().__init__()
        slope = _act_slope(activation)
        self.conv_dim, self.c_dim, self.repeat_num = conv_dim, c_dim, repeat_num
        self.num_channels, self.activation = num_channels, activation
        self.precision = L.PREC_F32
        self.wgrad_precision = None          # None: same as precision
        self.wgrad_side_stream = os.environ.get("FCD_SIDE_STREAM", "1") != "0"   # ops.SideWork
        # One weight-gradient launch per layer for the G step's two passes through the generator (ops.WgradPairing): the
        # reconstruction pass's backward (autograd runs it first) leaves its operand pairs, the translation pass's takes
        # them along.  FCD_MERGE_WGRAD=0: two launches per layer (A/B runs).
        self.merge_wgrad = os.environ.get("FCD_MERGE_WGRAD_G", os.environ.get("FCD_MERGE_WGRAD", "1")) != "0"
        # ... for launches of at most this many GFLOP per pass: a pair that waits for the second pass is weight-gradient
        # work the first pass's side stream does not do beside its norm backwards.  With 100 us launches (the benchmarked
        # size: 34-39 GFLOP) the saved slab reduction and launch outweigh that (-0.16 ms per step); with 3 ms launches
        # (the stress size: 1.1-1.2 TFLOP) they do not (+2 %, tools/stress_probe.py).
        self.merge_wgrad_max_gflop = float(os.environ.get("FCD_MERGE_WGRAD_MAX_GFLOP", "200"))
        self._pairing = ops.WgradPairing()
        cin = num_channels + c_dim
        mods: List[nn.Module] = []
        plan: List[_Layer] = []

        def add(kind, geom, conv, norm, act):
            plan.append(_Layer(kind, geom, conv, norm, act))

        c0 = ConvParams((conv_dim, cin, 7, 7), cin * 49)
        n0 = NormParams(conv_dim, False)
        mods += [c0, n0, _Slot()]
        add("conv", ConvGeom(conv_dim, cin, 7, 1, 3), c0, n0, slope)
        d = conv_dim
        for _ in range(2):
            cv = ConvParams((d * 2, d, 4, 4), d * 16)
            nm = NormParams(d * 2, False)
            mods += [cv, nm, _Slot()]
            add("conv", ConvGeom(d * 2, d, 4, 2, 1), cv, nm, slope)
            d *= 2
        for _ in range(repeat_num):
            blk = ResidualBlock(d, d)
            mods.append(blk)
            plan.append(_Layer("conv", ConvGeom(d, d, 3, 1, 1), blk.main[0], blk.main[1], slope))
            plan.append(_Layer("conv", ConvGeom(d, d, 3, 1, 1), blk.main[3], blk.main[4], 1.0, res_from=-2))
        for _ in range(2):
            # ConvTranspose2d(d, d/2, 4, 2, 1): weight [d][d/2][4][4]; torch's fan_in is dim 1 * k*k
            cv = ConvParams((d, d // 2, 4, 4), (d // 2) * 16)
            nm = NormParams(d // 2, False)
            mods += [cv, nm, _Slot()]
            add("convT", ConvGeom(d, d // 2, 4, 2, 1), cv, nm, slope)
            d //= 2
        cl = ConvParams((num_channels, d, 7, 7), d * 49)
        mods.append(cl)
        plan.append(_Layer("conv", ConvGeom(num_channels, d, 7, 1, 3), cl, None, 1.0))
        self.main = nn.Sequential(*mods)
        self._plan = plan
        self._wmax = WeightMaxima(ly.conv for ly in plan)
        self._groups = 1
        self._repeat_first = 0        # extra running-statistics updates with the first sample group (repeat_first_group)
        self._grad_hook = None        # parallel.BucketedReducer while a data-parallel G step runs
        self._grad_mode = True        # torch.is_grad_enabled() at the last forward call
        self._stats_epoch = 0         # bumped whenever a kernel rewrites the running statistics

    @contextlib.contextmanager
    def sample_groups(self, groups: int):
        """Inside this context a forward call on a batch of ``groups * B`` samples stands for
        ``groups`` consecutive reference calls on ``B`` samples each: InstanceNorm couples nothing
        across samples, so the outputs and gradients are those of the separate calls, and the tracked
        running statistics are updated once per group, in order, exactly as the separate calls would
        (``fcd_solver.py:316-357`` makes four generator calls per G step; the solver batches them in
        pairs so that every launch-bound kernel runs half as often)."""
        prev, self._groups = self._groups, int(groups)
        try:
            yield
        finally:
            self._groups = prev

    @contextlib.contextmanager
    def repeat_first_group(self, times: int = 1):
        """Inside this context (together with :meth:`sample_groups`) a forward call also stands for ``times`` EARLIER
        reference calls on its first sample group: the tracked running statistics receive ``times`` extra momentum
        updates with that group's batch statistics before the per-group updates.  The solver uses it when the D step
        and the G step of one training step share the generator call ``G(x_real, c_trg)`` (``fcd_solver.py:291, 316``:
        the same inputs through the same weights, computed twice by the reference): outputs are shared, and the
        running statistics still see every reference call, in the reference's order."""
        prev, self._repeat_first = self._repeat_first, int(times)
        try:
            yield
        finally:
            self._repeat_first = prev

    # parameters in plan order: (conv.weight, norm.weight, norm.bias)*
    def _flat_params(self):
        out = []
        for ly in self._plan:
            out.append(ly.conv.weight)
            if ly.norm is not None:
                out += [ly.norm.weight, ly.norm.bias]
        return out

    def forward(self, x, c):
        if not x.is_cuda:
            raise RuntimeError("fcd_amd.Generator runs on the MI355X HIP path only; move inputs and module to cuda")
        x = x.contiguous().float()
        c = c.contiguous().float().view(x.size(0), -1)
        self._grad_mode = torch.is_grad_enabled()     # (always False inside Function.forward)
        return _GeneratorFn.apply(self, x, c, *self._flat_params())


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: Generator, x, c, *params):
        L.require_gpu_tensor(x, "x")
        n, ch, h, w = x.shape
        if ch != mod.num_channels or c.shape[1] != mod.c_dim:
            raise RuntimeError(f"Generator: expected x with {mod.num_channels} bands and c with {mod.c_dim} labels")
        prec = mod.precision
        training = mod.training
        plan = mod._plan
        if n % mod._groups:
            raise RuntimeError(f"Generator: batch {n} is not a multiple of sample_groups({mod._groups})")
        # no backward will ask for the per-layer tensors: torch.no_grad(), or nothing requires grad
        # (needs_input_grad follows the inputs' requires_grad flags, not the grad mode)
        inference = not mod._grad_mode or not any(ctx.needs_input_grad)
        cp = plan[0].geom.IP
        acts = [ops.g_prologue(x, c, cp)]      # acts[i] = input of layer i
        # PREC_F16X2: max |acts[i]| as device scalars (written by the norm kernel that produces the activation, or by
        # an absmax launch where a conv that wants it has another producer); amx_ok[i]: already filled
        h2 = prec in L.F16_FAMILY
        amx = torch.zeros((len(plan) + 1, ops.AMAX_SLOTS), dtype=torch.float32, device=x.device) if h2 else None
        amx_ok = [False] * (len(plan) + 1)
        if h2:
            mod._wmax.refresh()

        def act_amax(i):
            if not amx_ok[i]:
                ops.absmax(acts[i], out=amx[i])
                amx_ok[i] = True
            return amx[i]

        saved = []                              # per layer: (z, mean, rstd, use_running)
        running = {}                            # (N, C) -> tracked norms whose running statistics this call updates
        pi = 0
        for li, ly in enumerate(plan):
            wt = params[pi]
            pi += 1
            a = acts[-1]
            g = ly.geom
            nm = ly.norm
            use_running = nm is not None and nm.track and not training
            fuse = nm is not None and not use_running      # statistics ride in the conv's epilogue
            if use_running and inference and ly.kind == "conv" and ly.act_slope <= 1.0:
                # eval-mode norm folded into the convolution (nothing is kept for a backward pass)
                # (PREC_F16X2: the folded weights are scaled and packed as two fp16 planes, max |input| comes from the
                # previous layer's epilogue and this layer's epilogue leaves max |output| for the next one)
                d = g.fwd_desc(n, a.shape[1], a.shape[2], precision=prec, act_slope=ly.act_slope)
                wp, shift = ly.conv.folded(g, d, nm, mod._stats_epoch)
                if h2:
                    if d.w_amax:
                        d.a_amax = act_amax(li).data_ptr()
                    ops.set_out_amax(d, amx[li + 1])
                    amx_ok[li + 1] = True
                res = acts[li + 1 + ly.res_from] if ly.res_from is not None else None
                acts.append(ops.conv_gather(d, a, wp, bias=shift, residual=res))
                saved.append(None)
                pi += 2                          # gamma, beta: folded into wp / shift
                continue
            if ly.kind == "conv":
                d = g.fwd_desc(n, a.shape[1], a.shape[2], precision=prec)
                wp = ly.conv.operand("fwd", g, d)
            else:
                d = g.adj_desc(n, a.shape[1], a.shape[2], precision=prec)
                wp = ly.conv.operand("adj", g, d)
            if h2 and d.w_amax:
                d.a_amax = act_amax(li).data_ptr()
            y = None
            if fuse:
                # convolution, statistics and the norm's output in one call (one launch less for the 32 x 32 layers)
                res = acts[li + 1 + ly.res_from] if ly.res_from is not None else None
                z, mean, rstd, varu, y = ops.conv_gather_stats(
                    d, a, wp, want_var=nm.track,
                    apply=(params[pi].detach(), params[pi + 1].detach(), ly.act_slope, res, amx[li + 1] if h2 else None))
            else:
                z = ops.conv_gather(d, a, wp)
            if nm is None:
                saved.append((z, None, None, False))
                acts.append(z)
                continue
            gamma, beta = params[pi], params[pi + 1]
            pi += 2
            if use_running:
                mean, rstd = ops.instnorm_stats_from_running(nm.running_mean, nm.running_var, n)
            elif nm.track:
                # one momentum update per reference call (sample group), in call order; the norms of equal
                # width go out in one launch after the layer loop
                running.setdefault(mean.shape, []).append((nm.running_mean, nm.running_var, mean, varu))
            if y is None:
                res = acts[li + 1 + ly.res_from] if ly.res_from is not None else None
                y = ops.instnorm_apply(z, mean, rstd, gamma.detach(), beta.detach(), ly.act_slope, residual=res,
                                       amax=amx[li + 1] if h2 else None)
            amx_ok[li + 1] = h2
            saved.append((z, mean, rstd, use_running))
            acts.append(y)
        for items in running.values():
            if mod._repeat_first:
                nfirst = n // mod._groups
                first = [(rm, rv, mean[:nfirst], varu[:nfirst]) for rm, rv, mean, varu in items]
                for _ in range(mod._repeat_first):
                    ops.instnorm_update_running_batch(first, groups=1)
            ops.instnorm_update_running_batch(items, groups=mod._groups)
        if running:
            mod._stats_epoch += 1      # raw-pointer writes do not bump tensor versions: folded() keys on this
        x_fake = ops.g_epilogue_fwd(acts[-1], x)
        ctx.mod = mod
        ctx.acts = acts
        ctx.saved_layers = saved
        ctx.x = x
        ctx.x_fake = x_fake
        ctx.prec = prec
        ctx.amx, ctx.amx_ok = amx, amx_ok
        return x_fake

    @staticmethod
    def backward(ctx, gout):
        mod, acts, saved, x, x_fake, prec = ctx.mod, ctx.acts, ctx.saved_layers, ctx.x, ctx.x_fake, ctx.prec
        prec_w = prec if mod.wgrad_precision is None else mod.wgrad_precision
        plan = mod._plan
        n = x.shape[0]
        need_x = ctx.needs_input_grad[1]
        gout = gout.contiguous()
        gx = torch.empty_like(x) if need_x else None
        cp_out = acts[-1].shape[3]
        g = ops.g_epilogue_bwd(gout, x, x_fake, cp_out, gx)   # grad wrt the last conv output (NHWC)
        # which params need grads (inputs: mod, x, c, *params)
        needs = ctx.needs_input_grad[3:]
        grads: List[Optional[torch.Tensor]] = [None] * len(needs)
        # data parallel: in the step's last pass through G every parameter gradient is final as soon as
        # its kernel is enqueued, and its arena span can go to the all-reduce beside the rest of the chain
        hook = mod._grad_hook
        final = hook is not None and any(needs) and hook.pass_begin()
        # index of each layer's first param
        pidx = []
        k = 0
        for ly in plan:
            pidx.append(k)
            k += 1 if ly.norm is None else 3
        pending_res = {}       # activation index -> residual gradient to add
        h2 = prec in L.F16_FAMILY or prec_w in L.F16_FAMILY
        amx, amx_ok = ctx.amx, ctx.amx_ok
        if h2 and amx is None:
            amx, amx_ok = torch.zeros((len(plan) + 1, ops.AMAX_SLOTS), dtype=torch.float32, device=x.device), [False] * (len(plan) + 1)
        gmx = torch.zeros((len(plan), ops.AMAX_SLOTS), dtype=torch.float32, device=x.device) if h2 else None   # max |dz| per layer

        def act_amax(i):
            if not amx_ok[i]:
                ops.absmax(acts[i], out=amx[i])
                amx_ok[i] = True
            return amx[i]

        deferred = []          # (s12, dgamma sink, dbeta sink) of the norms whose affine gradients are summed at the end
        side = ops.SideWork(x.device, mod.wgrad_side_stream)
        if hook is not None:
            hook.before_send = side.join
        for li in range(len(plan) - 1, -1, -1):
            ly = plan[li]
            z, mean, rstd, use_running = saved[li]
            a_in = acts[li]
            y = acts[li + 1]
            gm = ly.geom
            p0 = pidx[li]
            if ly.norm is not None:
                if ly.res_from is not None:
                    pending_res[li + 1 + ly.res_from] = g      # h_out = h_in + IN(..): pass g to h_in too
                want_aff = needs[p0 + 1] or needs[p0 + 2]
                sg, sb = grad_sink(ly.norm.weight, needs[p0 + 1]), grad_sink(ly.norm.bias, needs[p0 + 2])
                direct = want_aff and sg is not None and sb is not None
                # the ReLU mask is recomputed from z (beta given): the forward output is not read again
                if direct and not final:
                    # arena-backed gradients and nobody waiting for them before the pass ends: the sums over the
                    # images of all norms go out in one launch per width after the layer loop (same bits)
                    dz, s12 = ops.instnorm_bwd_deferred(g, z, mean, rstd, ly.norm.weight.detach(), ly.act_slope,
                                                        use_running, ly.norm.bias.detach(),
                                                        dz_amax=gmx[li] if h2 else None)
                    deferred.append((s12, sg, sb))
                    dgam = dbet = None
                else:
                    dz, dgam, dbet = ops.instnorm_bwd(g, None, z, mean, rstd,
                                                      ly.norm.weight.detach(), ly.act_slope, use_running,
                                                      want_affine=want_aff, dgamma=sg if direct else None,
                                                      dbeta=sb if direct else None, accumulate=direct,
                                                      beta=ly.norm.bias.detach(), dz_amax=gmx[li] if h2 else None)
                dz_ok = h2
                if not direct:
                    if needs[p0 + 1]:
                        grads[p0 + 1] = dgam
                    if needs[p0 + 2]:
                        grads[p0 + 2] = dbet
                if final:
                    hook.param_ready(ly.norm.bias)
                    hook.param_ready(ly.norm.weight)
            else:
                dz = g
                dz_ok = False
            def dz_amax():
                nonlocal dz_ok
                if not dz_ok:
                    ops.absmax(dz, out=gmx[li])
                    dz_ok = True
                return gmx[li]
            # weight gradient (PREC_F16X2: the kernels with a two-plane form want max |.| of both operands)
            h2_w = h2 and prec_w in L.F16_FAMILY
            if needs[p0]:
                sink = grad_sink(ly.conv.weight, True)     # arena-backed: accumulate into p.grad in place
                # (gsmall, xbig) of the weight gradient: a transposed convolution swaps the roles
                gs, xb = (dz, a_in) if ly.kind == "conv" else (a_in, dz)
                wd = gm.wgrad_desc(n, xb.shape[1], xb.shape[2], precision=prec_w, accumulate=sink is not None)
                mx = None
                if h2_w:
                    mx = (dz_amax(), act_amax(li)) if ly.kind == "conv" else (act_amax(li), dz_amax())
                    ops.set_wgrad_amax(wd, *mx)
                pairs = (mod.merge_wgrad and sink is not None and ops.conv_wgrad2_supported(wd)
                         and 2e-9 * wd.N * wd.GH * wd.GW * wd.O * wd.I * wd.KH * wd.KW <= mod.merge_wgrad_max_gflop)
                # the pair another pass of this backward call left for the layer rides in this launch; if none has and
                # this is not the step's last pass, this one leaves its own (Generator.merge_wgrad)
                extra = mod._pairing.take(li, lambda it: it[2] is sink and it[1].shape[1:] == xb.shape[1:]
                                          and (it[3] is None) == (mx is None)) if pairs else None
                if extra is not None:
                    dw = side.run(lambda: ops.conv_wgrad2(wd, gs, xb, extra[0], extra[1], dw=sink, amax2=extra[3]),
                                  gs, xb, *extra[:2], *(extra[3] or ()))
                elif pairs and not final:
                    mod._pairing.leave(li, (gs, xb, sink, mx),
                                       lambda wd=wd, gs=gs, xb=xb, sink=sink: ops.conv_wgrad(wd, gs, xb, dw=sink))
                    dw = None
                else:
                    dw = side.run(lambda: ops.conv_wgrad(wd, gs, xb, dw=sink), gs, xb)
                if sink is None:
                    grads[p0] = dw
            if final:
                # (a pair left behind that this pass could not take goes out before the gradient is declared final)
                if mod._pairing.pending(li):
                    side.run(lambda: mod._pairing.drain(li))
                hook.param_ready(ly.conv.weight)
            # data gradient
            if li == 0:
                if need_x:
                    d = gm.adj_desc(n, dz.shape[1], dz.shape[2], precision=prec)
                    wa = ly.conv.operand("adj", gm, d)
                    if h2 and d.w_amax:
                        d.a_amax = dz_amax().data_ptr()
                    g0 = ops.conv_gather(d, dz, wa)
                    ops.nhwc_to_nchw(g0, x.shape[1], out=gx, accumulate=True)
                break
            res = pending_res.pop(li, None)
            if ly.kind == "conv":
                d = gm.adj_desc(n, dz.shape[1], dz.shape[2], precision=prec)
                wa = ly.conv.operand("adj", gm, d)
                if h2 and d.w_amax:
                    d.a_amax = dz_amax().data_ptr()
                g = ops.conv_gather(d, dz, wa, residual=res)
            else:
                d = gm.fwd_desc(n, dz.shape[1], dz.shape[2], precision=prec)
                wa = ly.conv.operand("fwd", gm, d)
                if h2 and d.w_amax:
                    d.a_amax = dz_amax().data_ptr()
                g = ops.conv_gather(d, dz, wa, residual=res)
        if deferred:
            ops.instnorm_affine_batch(deferred, accumulate=True)
        side.join()
        if hook is not None:
            hook.before_send = None
        ctx.acts = ctx.saved_layers = None
        return (None, gx, None) + tuple(grads)
