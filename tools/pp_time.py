"""The 128 -> 64 up-sampling adjoint (four 2 x 2-tap phases, 64 output channels) at 32 and 16 samples: phase pairs on the
256 x 128 wave layout against one phase per workgroup on 256 x 64 tiles (FCD_NO_PP=1).   python tools/pp_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
g = ops.ConvGeom(128, 64, 4, 2, 1)
wd = torch.randn(128, 64, 4, 4, device=dev) / 45
wa = ops.absmax(wd)
for n in (32, 16):
    x = torch.randn(n, 64, 64, g.OP, device=dev)
    xa = ops.absmax(x)
    for env in ("1", None):
        if env: os.environ["FCD_NO_PP"] = env
        else: os.environ.pop("FCD_NO_PP", None)
        for prec in (3, 2):
            d = g.adj_desc(n, 64, 64, precision=prec)
            if prec == 3:
                ops.set_amax(d, xa, wa); wp = ops.pack_weight_adj_h2(wd, g, wa)
            else:
                wp = ops.pack_weight_adj_x3(wd, g)
            out = ops.conv_gather(d, x, wp)
            for _ in range(10): ops.conv_gather(d, x, wp, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): ops.conv_gather(d, x, wp, out=out)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            gf = 2.0 * n * 64 * 64 * 4 * 128 * 64 * 4 / 1e9
            print(f"n={n} prec {prec} {L.load().fcd_last_conv_kernel().decode()[22:]:60s} {us:7.1f} us  {gf / us / 1e3:6.1f} TF", flush=True)
