"""conv7_halo_x3_kernel (7x7, 64 -> <= 16 channels, BF16X3) against the exact-fp32 halo kernel (FCD_NO_HALO7X=1)
and torch in float64: results and time."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def run(O, n, h, w, x3, bias, iters=0):
    if x3: os.environ.pop("FCD_NO_HALO7X", None)
    else: os.environ["FCD_NO_HALO7X"] = "1"
    g = ops.ConvGeom(O, 64, 7, 1, 3)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, 64, h, w, generator=gen)
    wt = torch.randn(O, 64, 7, 7, generator=gen) / 56.0
    b = torch.randn(O, generator=gen) if bias else None
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    d = g.fwd_desc(n, h, w, precision=2, act_slope=0.2)
    d.splitk = 1
    wp = ops.pack_weight_fwd(wt.to(dev), g)
    bd = None if b is None else b.to(dev)
    out = ops.conv_gather(d, xd, wp, bias=bd)
    name = L.load().fcd_last_conv_kernel().decode()
    t = None
    if iters:
        for _ in range(10): ops.conv_gather(d, xd, wp, bias=bd, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.conv_gather(d, xd, wp, bias=bd, out=out)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=3), 0.2).permute(0, 2, 3, 1)
    return out.cpu(), ref, name, t


for (O, n, h, w, bias) in [(10, 2, 40, 24, False), (10, 3, 21, 45, True), (3, 1, 40, 33, True), (16, 2, 9, 70, False),
                           (12, 5, 64, 64, True), (10, 16, 128, 128, False), (10, 32, 128, 128, False), (10, 64, 128, 128, False)]:
    big = n >= 16
    a, ref, na, ta = run(O, n, h, w, False, bias, 30 if big else 0)
    b, _, nb, tb = run(O, n, h, w, True, bias, 30 if big else 0)
    e_a = float((a[..., :O].double() - ref).norm() / ref.norm()); e_b = float((b[..., :O].double() - ref).norm() / ref.norm())
    pad = float(b[..., O:].abs().max()) if b.shape[-1] > O else 0.0
    print((O, n, h, w, bias), na, "->", nb, f"err fp32 {e_a:.2e} x3 {e_b:.2e} max|a-b| {float((a-b).abs().max()):.2e} pad {pad}",
          f"fp32 {ta and round(ta,1)} us x3 {tb and round(tb,1)} us", flush=True)
