"""FCD_PREC_F16X2 (fp16 x 2 pieces, three MFMAs) gather kernels (A-halo and per-tap forms) against the BF16X3 ones and float64:
error, time.  Shapes: (O, I, N, H, W of the small side, form, operand scale[, kernel, stride, pad])."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def run(O, I, n, h, w, form, prec, scale=1.0, iters=0, k=3, st=1, pd=1):
    g = ops.ConvGeom(O, I, k, st, pd)
    gen = torch.Generator(device=dev).manual_seed(1)
    cin = I if form == "fwd" else O
    x = torch.randn(n, h, w, cin, device=dev, generator=gen) * abs(scale)
    wt = torch.randn(O, I, k, k, device=dev, generator=gen) / (k * I ** 0.5)
    if form == "fwd":       # (h, w) is the small side: the forward form reads the big one
        bh, bw = g.big_hw(h, w)
        x = torch.randn(n, bh, bw, cin, device=dev, generator=gen) * abs(scale)
        d = g.fwd_desc(n, bh, bw, precision=prec)
    else:
        d = g.adj_desc(n, h, w, precision=prec)
    if scale < 0:                 # a few outliers: max |x| / rms(x) = 4096 (what a per-tensor scale is weakest against)
        x.view(-1)[::1000003] *= 4096.0
    fmt = ops.weight_format(d)
    keep = []
    if fmt == 2:
        wa, xa = ops.absmax(wt), ops.absmax(x)
        keep = [wa, xa]
        ops.set_amax(d, xa, wa)
        wp = ops.pack_weight_fwd_h2(wt, g, wa) if form == "fwd" else ops.pack_weight_adj_h2(wt, g, wa)
    elif fmt == 1:
        wp = ops.pack_weight_fwd_x3(wt, g) if form == "fwd" else ops.pack_weight_adj_x3(wt, g)
    else:
        wp = ops.pack_weight_fwd(wt, g) if form == "fwd" else ops.pack_weight_adj(wt, g)
    out = ops.conv_gather(d, x, wp)
    name = L.load().fcd_last_conv_kernel().decode()
    t = None
    if iters:
        for _ in range(50): ops.conv_gather(d, x, wp, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.conv_gather(d, x, wp, out=out)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
    if n * h * w <= 40000:
        xc, wc = x.double().cpu().permute(0, 3, 1, 2), wt.double().cpu()
        ref = (F.conv2d(xc, wc, stride=st, padding=pd) if form == "fwd" else F.conv_transpose2d(xc, wc, stride=st, padding=pd)).permute(0, 2, 3, 1)
    else:
        ref = None
    return out.cpu(), ref, name, t


SHAPES = [(256, 256, 32, 32, 32, "fwd", 1.0), (256, 256, 32, 32, 32, "adj", 1e-7), (128, 96, 9, 64, 64, "fwd", 1.0),
          (256, 256, 16, 32, 32, "fwd", 300.0), (256, 256, 32, 32, 32, "fwd", -1.0), (256, 256, 32, 32, 32, "adj", -1e-6),
          # the generator's down / up-sampling layers (4x4, stride 2): per-tap gather form, 32 and 4 samples
          (128, 64, 32, 64, 64, "fwd", 1.0, 4, 2, 1), (256, 128, 32, 32, 32, "fwd", 1.0, 4, 2, 1),
          (256, 128, 32, 32, 32, "adj", 1e-5, 4, 2, 1), (128, 64, 32, 64, 64, "adj", 1.0, 4, 2, 1),
          (128, 64, 2, 64, 64, "fwd", 30.0, 4, 2, 1), (256, 128, 4, 32, 32, "adj", -1.0, 4, 2, 1), (128, 64, 2, 64, 64, "adj", 1.0, 4, 2, 1)]
for sh in SHAPES:
    O, I, n, h, w, form, sc = sh[:7]
    kw = dict(zip(("k", "st", "pd"), sh[7:]))
    a, ref, na, ta = run(O, I, n, h, w, form, 2, sc, 100, **kw)
    b, _, nb, tb = run(O, I, n, h, w, form, 3, sc, 100, **kw)
    c, _, nc, tc = run(O, I, n, h, w, form, 0, sc, 0, **kw)
    if ref is None: ref = c.double()
    e = lambda t: float((t.double() - ref).norm() / ref.norm())
    print(sh, nb[22:], f"err x3 {e(a):.2e} h2 {e(b):.2e} fp32 {e(c):.2e}  x3 {ta:.1f} us  h2 {tb:.1f} us", flush=True)
