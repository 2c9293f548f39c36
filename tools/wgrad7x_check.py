"""wgrad7_halo_x3_kernel (7x7 weight gradient, <= 16 output channels, BF16X3) against the exact-fp32 halo kernel
(FCD_NO_WGRAD7X=1) and torch in float64: results and time (slab reduction included)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def run(O, I, n, h, w, x3, iters=0):
    if x3: os.environ.pop("FCD_NO_WGRAD7X", None)
    else: os.environ["FCD_NO_WGRAD7X"] = "1"
    g = ops.ConvGeom(O, I, 7, 1, 3)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, I, h, w, generator=gen)
    gy = torch.randn(n, O, h, w, generator=gen)
    xd = torch.zeros(n, h, w, g.IP); xd[..., :I] = x.permute(0, 2, 3, 1)
    gd = torch.zeros(n, h, w, g.OP); gd[..., :O] = gy.permute(0, 2, 3, 1)
    xd, gd = xd.to(dev), gd.to(dev)
    d = g.wgrad_desc(n, h, w, precision=2)
    dw = ops.conv_wgrad(d, gd, xd)
    name = L.load().fcd_last_conv_kernel().decode()
    t = None
    if iters:
        for _ in range(10): ops.conv_wgrad(d, gd, xd, dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.conv_wgrad(d, gd, xd, dw)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
    ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, 7, 7), gy.double(), stride=1, padding=3) if n * h * w <= 70000 else None
    return dw.cpu(), ref, name, t


for (O, I, n, h, w) in [(10, 64, 2, 128, 128), (10, 32, 3, 40, 100), (16, 16, 1, 70, 64), (3, 48, 2, 33, 130), (12, 16, 5, 9, 7),
                        (10, 64, 16, 128, 128), (10, 64, 32, 128, 128)]:
    big = n >= 16
    a, ref, na, ta = run(O, I, n, h, w, False, 30 if big else 0)
    b, _, nb, tb = run(O, I, n, h, w, True, 30 if big else 0)
    if ref is None: ref = a.double()
    e_a = float((a.double() - ref).norm() / ref.norm()); e_b = float((b.double() - ref).norm() / ref.norm())
    print((O, I, n, h, w), na, "->", nb, f"err fp32 {e_a:.2e} x3 {e_b:.2e} max|a-b| {float((a-b).abs().max()):.2e}",
          f"fp32 {ta and round(ta,1)} us x3 {tb and round(tb,1)} us", flush=True)
