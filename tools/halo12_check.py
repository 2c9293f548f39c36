"""conv7_c12_halo_x3_kernel (7x7, 12 -> 64 channels, BF16X3) against the generic gather kernel (FCD_NO_HALO12=1)
and torch: results, fused statistics, time."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
torch.manual_seed(0)


def run(n, h, w, cin, halo, iters=0, stats=False):
    if halo: os.environ.pop("FCD_NO_HALO12", None)
    else: os.environ["FCD_NO_HALO12"] = "1"
    g = ops.ConvGeom(64, cin, 7, 1, 3)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(64, cin, 7, 7, generator=gen) / 23.0
    xd = torch.zeros(n, h, w, g.IP); xd[..., :cin] = x.permute(0, 2, 3, 1); xd = xd.to(dev)
    d = g.fwd_desc(n, h, w, precision=2)
    wp = ops.pack_weight_fwd(wt.to(dev), g)
    if stats:
        out, mean, rstd, varu = ops.conv_gather_stats(d, xd, wp, want_var=True)
    else:
        out = ops.conv_gather(d, xd, wp); mean = rstd = None
    name = L.load().fcd_last_conv_kernel().decode()
    t = None
    if iters:
        for _ in range(10): ops.conv_gather(d, xd, wp, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.conv_gather(d, xd, wp, out=out)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
    ref = F.conv2d(x.double(), wt.double(), padding=3).permute(0, 2, 3, 1)
    return out.cpu(), ref, mean, rstd, name, t


for (n, h, w, cin) in [(2, 40, 24, 11), (1, 9, 130, 10), (3, 64, 64, 11), (16, 128, 128, 11), (32, 128, 128, 11), (32, 128, 128, 10)]:
    big = n >= 16
    a, ref, ma, ra, na, ta = run(n, h, w, cin, False, 30 if big else 0, stats=True)
    b, _, mb, rb, nb, tb = run(n, h, w, cin, True, 30 if big else 0, stats=True)
    e_a = float((a.double() - ref).norm() / ref.norm()); e_b = float((b.double() - ref).norm() / ref.norm())
    m_ref = ref.mean((1, 2)); v_ref = ref.var((1, 2), unbiased=False)
    em = float((mb.cpu().double() - m_ref).abs().max()); er = float(((rb.cpu().double() - (v_ref + 1e-5).rsqrt()) / (v_ref + 1e-5).rsqrt()).abs().max())
    print((n, h, w, cin), nb, f"err old {e_a:.2e} new {e_b:.2e} max|a-b| {float((a-b).abs().max()):.2e} mean err {em:.1e} rstd rel err {er:.1e}",
          f"old {ta and round(ta,1)} us new {tb and round(tb,1)} us" , flush=True)
