"""FCD_PREC_F16X2 forms of the 7x7 halo kernels (fp32 weights, split inside the kernel; format 3) against their BF16X3 forms
and float64: error, time."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def run(O, I, n, h, w, form, prec, scale, iters=50):
    g = ops.ConvGeom(O, I, 7, 1, 3)
    gen = torch.Generator(device=dev).manual_seed(3)
    cin, cp = (I, g.IP) if form == "fwd" else (O, g.OP)
    x = torch.zeros(n, h, w, cp, device=dev)
    x[..., :cin] = torch.randn(n, h, w, cin, device=dev, generator=gen) * scale
    wt = torch.randn(O, I, 7, 7, device=dev, generator=gen) / (7 * I ** 0.5)
    d = g.fwd_desc(n, h, w, precision=prec) if form == "fwd" else g.adj_desc(n, h, w, precision=prec)
    fmt = ops.weight_format(d)
    keep = None
    if fmt == 3:
        keep = (ops.absmax(x), ops.absmax(wt))
        ops.set_amax(d, *keep)
    wp = ops.pack_weight_fwd(wt, g) if form == "fwd" else ops.pack_weight_adj(wt, g)
    out = ops.conv_gather(d, x, wp)
    name = L.load().fcd_last_conv_kernel().decode()
    for _ in range(10): ops.conv_gather(d, x, wp, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): ops.conv_gather(d, x, wp, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e3
    ref = None
    if n <= 4:
        xc, wc = x[..., :cin].double().cpu().permute(0, 3, 1, 2), wt.double().cpu()
        ref = (F.conv2d(xc, wc, padding=3) if form == "fwd" else F.conv_transpose2d(xc, wc, padding=3)).permute(0, 2, 3, 1)
    cout = O if form == "fwd" else I
    return out[..., :cout].cpu(), ref, name, t, fmt


for (O, I, n, h, w, form, sc) in [(64, 11, 2, 128, 128, "fwd", 1.0), (10, 64, 2, 128, 128, "adj", 1e-4), (10, 64, 2, 128, 128, "fwd", 3.0),
                                  (64, 11, 32, 128, 128, "fwd", 1.0), (10, 64, 32, 128, 128, "adj", 1.0), (10, 64, 32, 128, 128, "fwd", 1.0),
                                  (64, 11, 3, 40, 72, "fwd", 50.0)]:
    a, ref, na, ta, _ = run(O, I, n, h, w, form, 2, sc)
    b, _, nb, tb, fmt = run(O, I, n, h, w, form, 3, sc)
    if ref is None: ref = a.double()
    e = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
    print((O, I, n, h, w, form, sc), na, "->", nb, "fmt", fmt, f"err x3 {e(a):.2e} h2 {e(b):.2e}  x3 {ta:.1f} us  h2 {tb:.1f} us", flush=True)


def run_w(O, I, n, h, w, prec, sg, sx, iters=30):
    g = ops.ConvGeom(O, I, 7, 1, 3)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, I, h, w, generator=gen) * sx
    gy = torch.randn(n, O, h, w, generator=gen) * sg
    xd = torch.zeros(n, h, w, g.IP); xd[..., :I] = x.permute(0, 2, 3, 1)
    gd = torch.zeros(n, h, w, g.OP); gd[..., :O] = gy.permute(0, 2, 3, 1)
    xd, gd = xd.to(dev), gd.to(dev)
    d = g.wgrad_desc(n, h, w, precision=prec)
    if prec == 3: ops.set_wgrad_amax(d, ops.absmax(gd), ops.absmax(xd))
    dw = ops.conv_wgrad(d, gd, xd)
    name = L.load().fcd_last_conv_kernel().decode()
    for _ in range(5): ops.conv_wgrad(d, gd, xd, dw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): ops.conv_wgrad(d, gd, xd, dw)
    e1.record(); torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, 7, 7), gy.double(), stride=1, padding=3) if n * h * w <= 70000 else None
    return dw.cpu(), ref, name, e0.elapsed_time(e1) / iters * 1e3


for (O, I, n, h, w, sg, sx) in [(10, 64, 2, 128, 128, 1e-3, 1.0), (64, 11, 2, 128, 128, 1.0, 40.0), (16, 16, 1, 70, 64, 1.0, 1.0), (64, 11, 3, 40, 72, 1.0, 1.0),
                                (10, 64, 32, 128, 128, 1.0, 1.0), (64, 11, 32, 128, 128, 1.0, 1.0)]:
    a, ref, na, ta = run_w(O, I, n, h, w, 2, sg, sx)
    b, _, nb, tb = run_w(O, I, n, h, w, 3, sg, sx)
    if ref is None: ref = a.double()
    e = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
    print("wgrad", (O, I, n, h, w, sg, sx), na, "->", nb, f"err x3 {e(a):.2e} h2 {e(b):.2e}  x3 {ta:.1f} us  h2 {tb:.1f} us (with the slab reduction)", flush=True)
