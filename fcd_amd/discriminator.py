"""PatchGAN Discriminator of the Fixed-Point GAN on the HIP kernels.

Drop-in for the reference ``Discriminator`` (``models/fixed_point_gan.py:100-144``):
same constructor and ``forward(x) -> (out_src, out_cls, [feat])``, same state_dict
keys (``features.{i}.0.{weight,bias}``, ``main.{j}.0.{weight,bias}``,
``conv1.weight``, ``conv2.weight``).

Autograd structure: forward is one node; its backward is itself a differentiable
node (``_DBackwardFn``), because the WGAN-GP term (``fcd_solver.py:227-239``)
differentiates the input gradient again.  D has no norm layers and LeakyReLU has
zero second derivative, so the backward-of-backward consists of forward
convolutions of the incoming "grad of grad" and weight-gradient GEMMs against the
saved masked output gradients -- all explicit HIP launches, no autograd tracing
of elementary ops.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .arena import grad_sink
from .generator import ConvParams, WeightMaxima, _act_slope
from .ops import ConvGeom

class _State:
    """Fallback switch for ``input_grad_only()`` called without modules.  Process-wide on purpose, not
    thread-local: autograd runs a CUDA graph's backward nodes on its own worker thread, which must see
    the flag the calling thread set around ``torch.autograd.grad`` (the call blocks until that
    backward is done).  Prefer ``input_grad_only(D)``: the flag then lives on the module, and two
    solvers in one process do not share it."""
    input_only = False


_state = _State()


@contextlib.contextmanager
def input_grad_only(*modules):
    """Inside this context the backward of the given Discriminators (all Discriminators of the
    process when none is given) skips parameter gradients: the GP's
    ``autograd.grad(y, x, create_graph=True)`` only asks for d y / d x."""
    targets = modules if modules else (_state,)
    prev = [getattr(t, "input_only", False) for t in targets]
    for t in targets:
        t.input_only = True
    try:
        yield
    finally:
        for t, pv in zip(targets, prev):
            t.input_only = pv


def _input_only(mod) -> bool:
    return bool(getattr(mod, "input_only", False) or _state.input_only)


class _Acts(list):
    """The per-layer activations of one forward call; under the scaled-fp16 precisions also their maxima
    (``amx[i]`` = max |acts[i]| as AMAX_SLOTS partial maxima, written by the convolution that produced the tensor)."""
    amx = None


class _ConvAct(nn.Sequential):
    """``Sequential(Conv2d, act)``: index 0 holds the parameters (key ``*.0.weight``); with
    ``interm_non_act`` the reference builds ``Sequential(act, Conv2d)`` (key ``*.1.weight``)."""

    def __init__(self, cin, cout, act_first=False):
        conv = ConvParams((cout, cin, 4, 4), cin * 16, bias=True)
        super().__init__(*((nn.Identity(), conv) if act_first else (conv, nn.Identity())))
        self.conv_index = 1 if act_first else 0


class Discriminator(nn.Module):
    """Discriminator network with PatchGAN (HIP).  Reference: ``fixed_point_gan.py:100-144``."""

    def __init__(self, image_size=128, conv_dim=64, c_dim=5, repeat_num=6, num_channels=3, activation="lrelu",
                 n_feature_layers=4, interm_non_act=False, use_attention=False):
        super().__init__()
        if use_attention:
            # off by default (fcd_main.py:147); outside the hot path
            raise NotImplementedError("use_attention has no HIP path")
        # relu / lrelu: the backward masks come from the sign of the stored output (y > 0 <=> z > 0;
        # at z == 0 both torch and this path take the negative-side slope)
        # silu / mish: un-fused activations on stored pre-activations and a second-derivative sweep
        # in the double backward (_SmoothDiscriminatorFn below)
        self.slope = _act_slope(activation)
        self.smooth = self.slope > 1.0
        self.interm_non_act = bool(interm_non_act)
        self.image_size, self.c_dim, self.num_channels = image_size, c_dim, num_channels
        self.precision = L.PREC_F32
        self.wgrad_precision = None          # None: same as precision
        self.wgrad_side_stream = os.environ.get("FCD_SIDE_STREAM_D", os.environ.get("FCD_SIDE_STREAM", "1")) != "0"   # ops.SideWork
        layers = []
        geoms = []
        cin, cout = num_channels, conv_dim
        for _ in range(repeat_num):
            layers.append(_ConvAct(cin, cout, act_first=self.interm_non_act))
            geoms.append(ConvGeom(cout, cin, 4, 2, 1))
            cin, cout = cout, cout * 2
        ks = int(image_size / np.power(2, repeat_num))
        self.features = nn.ModuleList(layers[:n_feature_layers])
        rest = layers[n_feature_layers:]
        self.main = nn.Sequential(*rest) if len(rest) > 0 else nn.Identity()
        self.conv1 = ConvParams((1, cin, 3, 3), cin * 9)
        self.conv2 = ConvParams((c_dim, cin, ks, ks), cin * ks * ks)
        self._convs = [m[m.conv_index] for m in layers]
        self._geoms = geoms
        # interm_non_act (fixed_point_gan.py:108-111) builds every layer as act(inplace) -> conv.  The
        # in-place activation rewrites the previous layer's output, so every intermediate tensor --
        # the returned feature map included, as long as `main` is not empty -- ends up activated
        # exactly as in the default network; the one difference is that the LAST convolution's
        # output reaches the heads (and, when n_feature_layers >= repeat_num, the feature list)
        # without activation.  slope 1.0 = identity.
        self._slopes = [self.slope] * repeat_num
        if self.interm_non_act and repeat_num > 0:
            self._slopes[-1] = 1.0
        self._n_feat = min(n_feature_layers, len(layers))
        self._g1 = ConvGeom(1, cin, 3, 1, 1)
        self._g2 = ConvGeom(c_dim, cin, ks, 1, 0)
        self._grad_hook = None        # parallel.BucketedReducer while a data-parallel D step runs
        self.input_only = False       # input_grad_only(self): backward skips parameter gradients
        # One weight-gradient launch per layer for the D step's two backward passes (ops.conv_wgrad2): whichever of the
        # paired batch's backward and the gradient penalty's backward-of-backward autograd runs first leaves its operand
        # pair per layer here, the other takes it into its own launch; whatever is left when backward ends is flushed by
        # a callback.  FCD_MERGE_WGRAD=0: two launches per layer as before (A/B runs).
        self.merge_wgrad = os.environ.get("FCD_MERGE_WGRAD_D", os.environ.get("FCD_MERGE_WGRAD", "1")) != "0"
        self._pairing = ops.WgradPairing()
        self._wmax = WeightMaxima(self._convs)     # max |weight| of the six layers in one launch per optimizer step

    def _flat_params(self):
        out = []
        for cv in self._convs:
            out += [cv.weight, cv.bias]
        return out + [self.conv1.weight, self.conv2.weight]

    def _stash_wgrad(self, li, gz, u, sink) -> None:
        """Leave layer ``li``'s weight-gradient operands for a later backward pass of the same ``backward()`` call."""
        def alone():
            prec_w = self.precision if self.wgrad_precision is None else self.wgrad_precision
            wd = self._geoms[li].wgrad_desc(u.shape[0], u.shape[1], u.shape[2], precision=prec_w, accumulate=True)
            ops.conv_wgrad(wd, gz, u, dw=sink)
        self._pairing.leave(li, (gz, u, sink), alone)

    def _take_wgrad(self, li, wd, x, sink):
        """The operand pair an earlier pass of this backward call left for layer ``li``, if this pass can take it into its
        own launch (same gradient tensor, same geometry, a two-pair form for ``wd``), else None."""
        if sink is None or not ops.conv_wgrad2_supported(wd):
            return None
        return self._pairing.take(li, lambda it: it[2] is sink and tuple(it[1].shape[1:]) == tuple(x.shape[1:]))

    def _can_stash_wgrad(self, wd, sink, final: bool, scaled: bool) -> bool:
        return bool(self.merge_wgrad and sink is not None and not final and not scaled and ops.conv_wgrad2_supported(wd))

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("fcd_amd.Discriminator runs on the MI355X HIP path only")
        x = x.contiguous().float()
        fn = _SmoothDiscriminatorFn if self.smooth else _DiscriminatorFn
        out_src, out_cls, feat = fn.apply(self, x, *self._flat_params())
        return out_src, out_cls.view(out_cls.size(0), out_cls.size(1)), [feat]


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: Discriminator, x, *params):
        L.require_gpu_tensor(x, "x")
        n, ch, h, w = x.shape
        if ch != mod.num_channels:
            raise RuntimeError(f"Discriminator: expected {mod.num_channels} bands")
        prec = mod.precision
        acts = _Acts([ops.nchw_to_nhwc(x, mod._geoms[0].IP)])
        # Scaled-fp16 precisions (fp32x2: FCD_PREC_F16X2, fp16: FCD_PREC_F16): D has no InstanceNorm to record the maxima,
        # so every convolution leaves max |output| for the next one (FcdConvDesc.out_amax: from its epilogue, or from the
        # split-K reduction of the deep layers); only the input image needs a pass of its own
        h2 = prec in L.F16_FAMILY
        if h2:
            mod._wmax.refresh()
            acts.amx = torch.zeros((len(mod._convs) + 1, ops.AMAX_SLOTS), dtype=torch.float32, device=x.device)
            ops.absmax(acts[0], out=acts.amx[0])
        for li, (cv, g) in enumerate(zip(mod._convs, mod._geoms)):
            a = acts[-1]
            d = g.fwd_desc(n, a.shape[1], a.shape[2], precision=prec, act_slope=mod._slopes[li])
            wp = cv.operand("fwd", g, d)
            if h2:
                if d.w_amax:
                    d.a_amax = acts.amx[li].data_ptr()
                ops.set_out_amax(d, acts.amx[li + 1])
            acts.append(ops.conv_gather(d, a, wp, bias=params[2 * li + 1]))
        top = acts[-1]
        if top.shape[1] < mod._g2.k or top.shape[2] < mod._g2.k:
            raise RuntimeError("Discriminator: input smaller than image_size")
        out_src = ops.head_fwd(top, mod.conv1.packed("fwd", mod._g1), 1, 3, 1)
        out_cls = ops.head_fwd(top, mod.conv2.packed("fwd", mod._g2), mod.c_dim, mod._g2.k, 0)
        feat = ops.nhwc_to_nchw(acts[mod._n_feat], mod._geoms[mod._n_feat - 1].O)
        ctx.mod, ctx.acts, ctx.prec = mod, acts, prec
        ctx.set_materialize_grads(False)
        ctx.x_shape = x.shape
        return out_src, out_cls, feat

    @staticmethod
    def backward(ctx, g_src, g_cls, g_feat):
        mod = ctx.mod
        needs = ctx.needs_input_grad
        want_params = any(needs[2:]) and not _input_only(mod)
        outs = _DBackwardFn.apply(mod, ctx.acts, ctx.prec, needs[1], want_params, g_src, g_cls, g_feat,
                                  *mod._flat_params())
        gx = outs[0]
        gparams = outs[1:]
        res = [None, gx if needs[1] else None]
        for i, gp in enumerate(gparams):
            res.append(gp if (want_params and needs[2 + i]) else None)
        return tuple(res)


class _DBackwardFn(torch.autograd.Function):
    """(g_src, g_cls, g_feat, W...) -> (d x, d W...): D's first-order backward as a node."""

    @staticmethod
    def forward(ctx, mod: Discriminator, acts, prec, want_x, want_params, g_src, g_cls, g_feat, *params):
        n = acts[0].shape[0]
        slopes = mod._slopes
        slope = slopes[-1]               # activation in front of the heads
        prec_w = prec if mod.wgrad_precision is None else mod.wgrad_precision
        top = acts[-1]
        nl = len(mod._convs)
        dev = top.device
        grads: List[Optional[torch.Tensor]] = [None] * (2 * nl + 2)
        # Only the feature map is used downstream (feature-matching loss of a frozen D,
        # fcd_solver.py:361-385): start at the feature layer instead of pushing zeros through the
        # layers above it.
        only_feat = g_src is None and g_cls is None and g_feat is not None and not want_params
        # data parallel: see _GeneratorFn.backward (parameters become final heads first, then layers downwards)
        hook = mod._grad_hook
        final = hook is not None and want_params and hook.pass_begin()
        side = ops.SideWork(dev, mod.wgrad_side_stream)      # weight / bias gradients beside the data-gradient chain
        if hook is not None:
            hook.before_send = side.join
        # heads -> masked gradient at the top activation
        gz = None if only_feat else torch.empty_like(top)
        first = True
        if g_src is not None:
            g_src = g_src.contiguous()
            ops.head_dgrad(g_src, mod.conv1.packed("fwd", mod._g1), gz, top, slope, False, 3, 1)
            first = False
            if want_params:
                sink = grad_sink(mod.conv1.weight, True)
                dw = ops.head_wgrad(g_src, top, mod._g1.I, 3, 1, dw=sink, accumulate=sink is not None)
                grads[2 * nl] = None if sink is not None else dw
        if g_cls is not None:
            g_cls4 = g_cls.contiguous().view(n, mod.c_dim, 1, 1)
            ops.head_dgrad(g_cls4, mod.conv2.packed("fwd", mod._g2), gz, top, slope, not first, mod._g2.k, 0)
            first = False
            if want_params:
                sink = grad_sink(mod.conv2.weight, True)
                dw = ops.head_wgrad(g_cls4, top, mod._g2.I, mod._g2.k, 0, dw=sink, accumulate=sink is not None)
                grads[2 * nl + 1] = None if sink is not None else dw
        if first and gz is not None:
            gz.zero_()
        if final:
            hook.param_ready(mod.conv2.weight)
            hook.param_ready(mod.conv1.weight)
        gzs = [None] * nl          # gzs[l] = masked gradient at layer l's pre-activation
        gx = None
        # scaled-fp16 precisions: gamx[l] = max |gzs[l]| (the heads' gradient by a pass of its own, the others from the
        # epilogue of the adjoint convolution that writes them)
        h2 = prec in L.F16_FAMILY and getattr(acts, "amx", None) is not None
        gamx = torch.zeros((nl, ops.AMAX_SLOTS), dtype=torch.float32, device=dev) if h2 else None
        if h2 and gz is not None:
            ops.absmax(gz, out=gamx[nl - 1])
        for li in range(mod._n_feat - 1 if only_feat else nl - 1, -1, -1):
            if g_feat is not None and li == mod._n_feat - 1:
                # gradient arriving at the exposed feature map: mask it and add
                gf = ops.nchw_to_nhwc(g_feat.contiguous(), acts[li + 1].shape[3])
                L.check(L.load().fcd_lrelu_bwd(L.ptr(gf), L.ptr(acts[li + 1]), L.ptr(gf), gf.numel(), slopes[li],
                                               L.stream()), "lrelu_bwd")
                gz = gf if gz is None else gz + gf
                if h2:
                    gamx[li].zero_()
                    ops.absmax(gz, out=gamx[li])
            gzs[li] = gz
            g = mod._geoms[li]
            a_in = acts[li]
            if want_params:
                cv = mod._convs[li]
                sink = grad_sink(cv.weight, True)
                wd = g.wgrad_desc(n, a_in.shape[1], a_in.shape[2], precision=prec_w, accumulate=sink is not None)
                if h2 and prec_w in L.F16_FAMILY:
                    ops.set_wgrad_amax(wd, gamx[li], acts.amx[li])
                bsink = grad_sink(cv.bias, gz.shape[3] == g.O)
                # an operand pair the gradient penalty's pass left for this layer rides in the same launch; if that pass
                # is still to come, this one leaves its pair instead (Discriminator.merge_wgrad)
                scaled = h2 and prec_w in L.F16_FAMILY
                extra = None if scaled else mod._take_wgrad(li, wd, a_in, sink)
                if extra is not None:
                    wfn = lambda: ops.conv_wgrad2(wd, gz, a_in, extra[0], extra[1], dw=sink)
                elif mod._can_stash_wgrad(wd, sink, final, scaled):
                    mod._stash_wgrad(li, gz, a_in, sink)
                    wfn = lambda: None
                else:
                    wfn = lambda: ops.conv_wgrad(wd, gz, a_in, dw=sink)
                # (results that come back as fresh tensors are sliced below, on this stream: those stay on it)
                run = side.run if sink is not None and bsink is not None else (lambda fn, *keep: fn())
                dw, bs = run(lambda: (wfn(),
                                      ops.colsum(gz.numel() // gz.shape[3], gz, gz.shape[3], out=bsink,
                                                 accumulate=bsink is not None)), gz, a_in,
                             *(extra[:2] if extra is not None else ()))
                grads[2 * li] = None if sink is not None else dw
                if bsink is None:
                    grads[2 * li + 1] = bs if gz.shape[3] == g.O else bs[: g.O].contiguous()
                if final:
                    if mod._pairing.pending(li):      # (a pair left behind that this pass could not take)
                        side.run(lambda: mod._pairing.drain(li))
                    hook.param_ready(cv.bias)
                    hook.param_ready(cv.weight)
            if li == 0:
                if want_x:
                    d = g.adj_desc(n, gz.shape[1], gz.shape[2], precision=prec)
                    wp = mod._convs[0].operand("adj", g, d)
                    if h2 and d.w_amax:
                        d.a_amax = gamx[0].data_ptr()
                    g0 = ops.conv_gather(d, gz, wp)
                    gx = ops.nhwc_to_nchw(g0, mod.num_channels)
                break
            d = g.adj_desc(n, gz.shape[1], gz.shape[2], precision=prec, mask_slope=slopes[li - 1])
            wp = mod._convs[li].operand("adj", g, d)
            if h2:
                if d.w_amax:
                    d.a_amax = gamx[li].data_ptr()
                ops.set_out_amax(d, gamx[li - 1])
            gz = ops.conv_gather(d, gz, wp, mask=a_in)
        side.join()
        if hook is not None:
            hook.before_send = None
        ctx.mod, ctx.acts, ctx.prec, ctx.gzs = mod, acts, prec, gzs
        ctx.gamx = gamx
        ctx.g_src, ctx.g_cls = g_src, g_cls
        ctx.set_materialize_grads(False)
        if gx is None:
            gx = torch.zeros((n, mod.num_channels) + tuple(acts[0].shape[1:3]), device=dev)
        return (gx,) + tuple(grads)

    @staticmethod
    def backward(ctx, gg_x, *gg_params):
        """Backward-of-backward: only d(dx)/dW is supported (what the gradient penalty needs)."""
        if any(t is not None for t in gg_params):
            raise NotImplementedError("differentiating through D's parameter gradients is not supported")
        mod, acts, prec, gzs = ctx.mod, ctx.acts, ctx.prec, ctx.gzs
        prec_w = prec if mod.wgrad_precision is None else mod.wgrad_precision
        nl = len(mod._convs)
        # inputs: mod, acts, prec, want_x, want_params, g_src, g_cls, g_feat, *params
        needs = ctx.needs_input_grad[8:]
        out: List[Optional[torch.Tensor]] = [None] * len(needs)
        if gg_x is None:
            return (None,) * 8 + tuple(out)
        n = acts[0].shape[0]
        # data parallel: this pass walks D upwards; when it is the step's last one (autograd usually runs it
        # first) the spans simply complete in forward order
        hook = mod._grad_hook
        final = hook is not None and any(needs) and hook.pass_begin()
        side = ops.SideWork(acts[0].device, mod.wgrad_side_stream)
        if hook is not None:
            hook.before_send = side.join
        u = ops.nchw_to_nhwc(gg_x.contiguous(), mod._geoms[0].IP)
        gamx = ctx.gamx
        h2 = prec in L.F16_FAMILY and gamx is not None
        uamx = None
        if h2:
            uamx = torch.zeros((nl + 1, ops.AMAX_SLOTS), dtype=torch.float32, device=u.device)
            ops.absmax(u, out=uamx[0])
        for li in range(nl):
            g = mod._geoms[li]
            if needs[2 * li]:
                if gzs[li] is None:
                    raise NotImplementedError("double backward through a feature-only backward of D")
                sink = grad_sink(mod._convs[li].weight, True)
                wd = g.wgrad_desc(n, u.shape[1], u.shape[2], precision=prec_w, accumulate=sink is not None)
                if h2 and prec_w in L.F16_FAMILY:
                    ops.set_wgrad_amax(wd, gamx[li], uamx[li])
                scaled = h2 and prec_w in L.F16_FAMILY
                extra = None if scaled else mod._take_wgrad(li, wd, u, sink)
                if extra is not None:
                    # the paired batch's pass left its operand pair for this layer: one launch, one pass over the gradient
                    dw = side.run(lambda wd=wd, gz=gzs[li], u=u, sink=sink, e=extra: ops.conv_wgrad2(wd, gz, u, e[0], e[1], dw=sink),
                                  gzs[li], u, extra[0], extra[1])
                elif mod._can_stash_wgrad(wd, sink, final, scaled):
                    # another pass of this backward call may produce the same gradient: it takes this pair along
                    mod._stash_wgrad(li, gzs[li], u, sink)
                    dw = None
                else:
                    dw = side.run(lambda wd=wd, gz=gzs[li], u=u, sink=sink: ops.conv_wgrad(wd, gz, u, dw=sink), gzs[li], u)
                out[2 * li] = None if sink is not None else dw
            if needs[2 * li + 1] and grad_sink(mod._convs[li].bias, True) is None:
                out[2 * li + 1] = torch.zeros_like(mod._convs[li].bias)
            if final:
                if mod._pairing.pending(li):          # (a pair left behind that this pass could not take)
                    side.run(lambda: mod._pairing.drain(li))
                hook.param_ready(mod._convs[li].weight)
                hook.param_ready(mod._convs[li].bias)
            d = g.fwd_desc(n, u.shape[1], u.shape[2], precision=prec, mask_slope=mod._slopes[li])
            wp = mod._convs[li].operand("fwd", g, d)
            if h2:
                if d.w_amax:
                    d.a_amax = uamx[li].data_ptr()
                ops.set_out_amax(d, uamx[li + 1])
            u = ops.conv_gather(d, u, wp, mask=acts[li + 1])
        if needs[2 * nl] and ctx.g_src is not None:
            sink = grad_sink(mod.conv1.weight, True)
            dw = ops.head_wgrad(ctx.g_src, u, mod._g1.I, 3, 1, dw=sink, accumulate=sink is not None)
            out[2 * nl] = None if sink is not None else dw
        if needs[2 * nl + 1] and ctx.g_cls is not None:
            sink = grad_sink(mod.conv2.weight, True)
            dw = ops.head_wgrad(ctx.g_cls.contiguous().view(n, mod.c_dim, 1, 1), u, mod._g2.I, mod._g2.k, 0, dw=sink,
                                accumulate=sink is not None)
            out[2 * nl + 1] = None if sink is not None else dw
        if final:
            hook.param_ready(mod.conv1.weight)
            hook.param_ready(mod.conv2.weight)
        # d/d g_src, d/d g_cls (linear) are not needed by the GP: grad_outputs is a constant
        side.join()
        if hook is not None:
            hook.before_send = None
        ctx.gzs = ctx.gamx = None
        return (None,) * 8 + tuple(out)


# --------------------------------------------------------------------------- #
# SiLU / Mish discriminator (fixed_point_gan.py:13-16)
# --------------------------------------------------------------------------- #
# With z_l = W_l * a_{l-1} + b_l, a_l = act(z_l), the first backward is
#     gz_l = ga_l . act'(z_l),   ga_{l-1} = W_l^T gz_l,   gx = ga_{-1}.
# act'' != 0, so the backward of that backward (the gradient penalty's) has, besides the forward
# convolutions of the incoming cotangent u (ubar_l = W_l u_{l-1}, u_l = ubar_l . act'(z_l)) and the
# weight gradients gz_l (x) u_{l-1}, a second sweep: q_l = ubar_l . ga_l . act''(z_l) is a cotangent
# of z_l itself and runs down the forward network like an ordinary backward pass
#     r_l = q_l + (W_{l+1}^T r_{l+1}) . act'(z_l),   dW_l += r_l (x) a_{l-1},   db_l += sum r_l.
# The activations are separate elementwise launches here (fcd_act_fwd / _bwd / _bwd2): the option is
# off by default and D is 7 % of the step.
class _SmoothDiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: Discriminator, x, *params):
        L.require_gpu_tensor(x, "x")
        n, ch, h, w = x.shape
        if ch != mod.num_channels:
            raise RuntimeError(f"Discriminator: expected {mod.num_channels} bands")
        prec = mod.precision
        acts = [ops.nchw_to_nhwc(x, mod._geoms[0].IP)]
        zs = []
        for li, (cv, g) in enumerate(zip(mod._convs, mod._geoms)):
            a = acts[-1]
            d = g.fwd_desc(n, a.shape[1], a.shape[2], precision=prec)
            z = ops.conv_gather(d, a, cv.operand("fwd", g, d), bias=params[2 * li + 1])
            zs.append(z)
            acts.append(z if mod._slopes[li] == 1.0 else ops.act_fwd(z, mod._slopes[li]))
        top = acts[-1]
        if top.shape[1] < mod._g2.k or top.shape[2] < mod._g2.k:
            raise RuntimeError("Discriminator: input smaller than image_size")
        out_src = ops.head_fwd(top, mod.conv1.packed("fwd", mod._g1), 1, 3, 1)
        out_cls = ops.head_fwd(top, mod.conv2.packed("fwd", mod._g2), mod.c_dim, mod._g2.k, 0)
        feat = ops.nhwc_to_nchw(acts[mod._n_feat], mod._geoms[mod._n_feat - 1].O)
        ctx.mod, ctx.acts, ctx.zs, ctx.prec = mod, acts, zs, prec
        ctx.set_materialize_grads(False)
        return out_src, out_cls, feat

    @staticmethod
    def backward(ctx, g_src, g_cls, g_feat):
        mod = ctx.mod
        needs = ctx.needs_input_grad
        want_params = any(needs[2:]) and not _input_only(mod)
        outs = _SmoothDBackwardFn.apply(mod, ctx.acts, ctx.zs, ctx.prec, needs[1], want_params, g_src, g_cls, g_feat,
                                        *mod._flat_params())
        res = [None, outs[0] if needs[1] else None]
        for i, gp in enumerate(outs[1:]):
            res.append(gp if (want_params and needs[2 + i]) else None)
        return tuple(res)


def _add_grad(slot, new):
    return new if slot is None else slot + new


class _SmoothDBackwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: Discriminator, acts, zs, prec, want_x, want_params, g_src, g_cls, g_feat, *params):
        n = acts[0].shape[0]
        codes = mod._slopes
        prec_w = prec if mod.wgrad_precision is None else mod.wgrad_precision
        top = acts[-1]
        nl = len(mod._convs)
        grads: List[Optional[torch.Tensor]] = [None] * (2 * nl + 2)
        if mod._grad_hook is not None and want_params:
            mod._grad_hook.pass_begin()       # counted; spans of the SiLU / Mish path go out after backward
        ga = torch.empty_like(top)            # cotangent of a_top (no activation factor yet)
        first = True
        if g_src is not None:
            g_src = g_src.contiguous()
            ops.head_dgrad(g_src, mod.conv1.packed("fwd", mod._g1), ga, top, 1.0, False, 3, 1)
            first = False
            if want_params:
                sink = grad_sink(mod.conv1.weight, True)
                dw = ops.head_wgrad(g_src, top, mod._g1.I, 3, 1, dw=sink, accumulate=sink is not None)
                grads[2 * nl] = None if sink is not None else dw
        if g_cls is not None:
            g_cls4 = g_cls.contiguous().view(n, mod.c_dim, 1, 1)
            ops.head_dgrad(g_cls4, mod.conv2.packed("fwd", mod._g2), ga, top, 1.0, not first, mod._g2.k, 0)
            first = False
            if want_params:
                sink = grad_sink(mod.conv2.weight, True)
                dw = ops.head_wgrad(g_cls4, top, mod._g2.I, mod._g2.k, 0, dw=sink, accumulate=sink is not None)
                grads[2 * nl + 1] = None if sink is not None else dw
        if first:
            ga.zero_()
        gas, gzs = [None] * nl, [None] * nl
        gx = None
        for li in range(nl - 1, -1, -1):
            if g_feat is not None and li == mod._n_feat - 1:
                ga = ga + ops.nchw_to_nhwc(g_feat.contiguous(), acts[li + 1].shape[3])
            gas[li] = ga
            gz = ga if codes[li] == 1.0 else ops.act_bwd(ga, zs[li], codes[li])
            gzs[li] = gz
            g = mod._geoms[li]
            a_in = acts[li]
            if want_params:
                cv = mod._convs[li]
                sink = grad_sink(cv.weight, True)
                wd = g.wgrad_desc(n, a_in.shape[1], a_in.shape[2], precision=prec_w, accumulate=sink is not None)
                dw = ops.conv_wgrad(wd, gz, a_in, dw=sink)
                grads[2 * li] = None if sink is not None else dw
                bsink = grad_sink(cv.bias, gz.shape[3] == g.O)
                bs = ops.colsum(gz.numel() // gz.shape[3], gz, gz.shape[3], out=bsink, accumulate=bsink is not None)
                if bsink is None:
                    grads[2 * li + 1] = bs if gz.shape[3] == g.O else bs[: g.O].contiguous()
            if li == 0 and not want_x:
                break
            d = g.adj_desc(n, gz.shape[1], gz.shape[2], precision=prec)
            ga = ops.conv_gather(d, gz, mod._convs[li].operand("adj", g, d))
            if li == 0:
                gx = ops.nhwc_to_nchw(ga, mod.num_channels)
        ctx.mod, ctx.acts, ctx.zs, ctx.prec, ctx.gas, ctx.gzs = mod, acts, zs, prec, gas, gzs
        ctx.g_src, ctx.g_cls = g_src, g_cls
        ctx.set_materialize_grads(False)
        if gx is None:
            gx = torch.zeros((n, mod.num_channels) + tuple(acts[0].shape[1:3]), device=top.device)
        return (gx,) + tuple(grads)

    @staticmethod
    def backward(ctx, gg_x, *gg_params):
        if any(t is not None for t in gg_params):
            raise NotImplementedError("differentiating through D's parameter gradients is not supported")
        mod, acts, zs, prec, gas, gzs = ctx.mod, ctx.acts, ctx.zs, ctx.prec, ctx.gas, ctx.gzs
        prec_w = prec if mod.wgrad_precision is None else mod.wgrad_precision
        codes = mod._slopes
        nl = len(mod._convs)
        # inputs: mod, acts, zs, prec, want_x, want_params, g_src, g_cls, g_feat, *params
        needs = ctx.needs_input_grad[9:]
        out: List[Optional[torch.Tensor]] = [None] * len(needs)
        if gg_x is None:
            return (None,) * 9 + tuple(out)
        n = acts[0].shape[0]
        if mod._grad_hook is not None and any(needs):
            mod._grad_hook.pass_begin()

        def add_w(li, gz, a):
            sink = grad_sink(mod._convs[li].weight, True)
            g = mod._geoms[li]
            wd = g.wgrad_desc(n, a.shape[1], a.shape[2], precision=prec_w, accumulate=sink is not None)
            dw = ops.conv_wgrad(wd, gz, a, dw=sink)
            if sink is None:
                out[2 * li] = _add_grad(out[2 * li], dw)

        # sweep 1 (up): cotangents of the first backward's gz_l / ga_l
        u = ops.nchw_to_nhwc(gg_x.contiguous(), mod._geoms[0].IP)
        qs = [None] * nl
        for li in range(nl):
            g = mod._geoms[li]
            if needs[2 * li]:
                add_w(li, gzs[li], u)
            d = g.fwd_desc(n, u.shape[1], u.shape[2], precision=prec)
            ubar = ops.conv_gather(d, u, mod._convs[li].operand("fwd", g, d))
            if codes[li] == 1.0:
                u = ubar
            else:
                u, qs[li] = ops.act_bwd2(ubar, gas[li], zs[li], codes[li])
        if needs[2 * nl] and ctx.g_src is not None:
            sink = grad_sink(mod.conv1.weight, True)
            dw = ops.head_wgrad(ctx.g_src, u, mod._g1.I, 3, 1, dw=sink, accumulate=sink is not None)
            out[2 * nl] = None if sink is not None else dw
        if needs[2 * nl + 1] and ctx.g_cls is not None:
            sink = grad_sink(mod.conv2.weight, True)
            dw = ops.head_wgrad(ctx.g_cls.contiguous().view(n, mod.c_dim, 1, 1), u, mod._g2.I, mod._g2.k, 0, dw=sink,
                                accumulate=sink is not None)
            out[2 * nl + 1] = None if sink is not None else dw
        # sweep 2 (down): the cotangents q_l of the pre-activations through the forward network
        r = None                      # cotangent of a_l arriving from the layer above
        for li in range(nl - 1, -1, -1):
            if r is None:
                r = qs[li]
            elif codes[li] == 1.0:
                pass                  # a_l == z_l
            elif qs[li] is not None:
                r = ops.act_bwd(r, zs[li], codes[li], out=qs[li], accumulate=True)
            else:
                r = ops.act_bwd(r, zs[li], codes[li])
            if r is None:
                continue
            g = mod._geoms[li]
            if needs[2 * li]:
                add_w(li, r, acts[li])
            if needs[2 * li + 1]:
                cv = mod._convs[li]
                bsink = grad_sink(cv.bias, r.shape[3] == g.O)
                bs = ops.colsum(r.numel() // r.shape[3], r, r.shape[3], out=bsink, accumulate=bsink is not None)
                if bsink is None:
                    out[2 * li + 1] = bs if r.shape[3] == g.O else bs[: g.O].contiguous()
            if li > 0:
                d = g.adj_desc(n, r.shape[1], r.shape[2], precision=prec)
                r = ops.conv_gather(d, r, mod._convs[li].operand("adj", g, d))
        for li in range(nl):          # parameters that received nothing still get a defined gradient
            if needs[2 * li + 1] and out[2 * li + 1] is None and grad_sink(mod._convs[li].bias, True) is None:
                out[2 * li + 1] = torch.zeros_like(mod._convs[li].bias)
        ctx.gas = ctx.gzs = None
        return (None,) * 9 + tuple(out)
