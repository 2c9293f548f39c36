"""wgrad7_c12_halo_x3_kernel (7x7 weight gradient, 12 physical input channels, BF16X3) against the generic split-once
kernel (FCD_NO_WGRAD12=1): results and time (slab reduction included)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def run(O, I, n, h, w, halo, iters=0):
    if halo: os.environ.pop("FCD_NO_WGRAD12", None)
    else: os.environ["FCD_NO_WGRAD12"] = "1"
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


for (O, I, n, h, w) in [(64, 11, 2, 40, 24), (64, 10, 1, 9, 130), (40, 12, 3, 64, 64), (17, 11, 2, 33, 70), (64, 11, 4, 128, 128),
                        (64, 11, 16, 128, 128), (64, 11, 32, 128, 128)]:
    big = n >= 16
    a, ref, na, ta = run(O, I, n, h, w, False, 30 if big else 0)
    b, _, nb, tb = run(O, I, n, h, w, True, 30 if big else 0)
    if ref is None: ref = a.double()
    e_a = float((a.double() - ref).norm() / ref.norm()); e_b = float((b.double() - ref).norm() / ref.norm())
    print((O, I, n, h, w), na[:28], "->", nb, f"err generic {e_a:.2e} halo {e_b:.2e} max|a-b| {float((a-b).abs().max()):.2e}",
          f"generic {ta and round(ta,1)} us halo {tb and round(tb,1)} us", flush=True)
