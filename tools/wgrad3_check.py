"""wgrad3_halo_x3_kernel (3x3 weight gradient on 32-wide images, BF16X3, all nine taps per workgroup) against the generic
split-once kernel (FCD_NO_WGRAD3=1) and torch in float64: results and time (slab reduction included)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def run(O, I, n, h, halo, iters=0, prec=2, gscale=1.0):
    if halo: os.environ.pop("FCD_NO_WGRAD3", None)
    else: os.environ["FCD_NO_WGRAD3"] = "1"
    g = ops.ConvGeom(O, I, 3, 1, 1)
    gen = torch.Generator(device=dev).manual_seed(1)
    xd = torch.randn(n, h, 32, g.IP, device=dev, generator=gen)
    gd = torch.randn(n, h, 32, g.OP, device=dev, generator=gen) * gscale
    d = g.wgrad_desc(n, h, 32, precision=prec)
    keep = (ops.absmax(gd), ops.absmax(xd))
    if prec == 3: ops.set_wgrad_amax(d, *keep)
    dw = ops.conv_wgrad(d, gd, xd)
    name = L.load().fcd_last_conv_kernel().decode()
    t = None
    if iters:
        for _ in range(30): ops.conv_wgrad(d, gd, xd, dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.conv_wgrad(d, gd, xd, dw)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
    ref = None
    if n * h <= 300:
        ref = torch.nn.grad.conv2d_weight(xd[..., :I].permute(0, 3, 1, 2).double().cpu(), (O, I, 3, 3),
                                          gd[..., :O].permute(0, 3, 1, 2).double().cpu(), stride=1, padding=1)
    return dw.cpu(), ref, name, t


for (O, I, n, h) in [(64, 64, 6, 5), (128, 64, 4, 32), (256, 192, 3, 7), (256, 256, 16, 32), (256, 256, 32, 32), (256, 256, 64, 32)]:
    big = n * h > 300
    a, ref, na, ta = run(O, I, n, h, False, 50 if big else 0)
    b, _, nb, tb = run(O, I, n, h, True, 50 if big else 0)
    c, _, nc, tc = run(O, I, n, h, True, 50 if big else 0, prec=3, gscale=1e-6)
    c = c * 1e6
    if ref is None: ref = a.double()
    print("   fp16x2:", nc, f"err {float((c.double() - ref).norm() / ref.norm()):.2e}", f"{tc and round(tc, 1)} us", flush=True)
    e_a = float((a.double() - ref).norm() / ref.norm()); e_b = float((b.double() - ref).norm() / ref.norm())
    print((O, I, n, h), na[:30], "->", nb[:30], f"err generic {e_a:.2e} halo {e_b:.2e} max|a-b| {float((a-b).abs().max()):.2e}",
          f"generic {ta and round(ta,1)} us halo {tb and round(tb,1)} us", flush=True)
