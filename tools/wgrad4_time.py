"""Weight gradient of the generator's four 4x4 stride-2 layers at 32 samples: the halo kernel's MODE 1 against the
generic split-once kernel (FCD_NO_WGRAD4=1), fp32x2.   python tools/wgrad4_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
for (O, I, n, h) in ((128, 64, 32, 64), (256, 128, 32, 32)):
    g = ops.ConvGeom(O, I, 4, 2, 1)
    big = torch.randn(n, 2 * h, 2 * h, g.IP, device=dev)
    small = torch.randn(n, h, h, g.OP, device=dev)
    ga, xa = ops.absmax(small), ops.absmax(big)
    for env in ("1", None):
        if env: os.environ["FCD_NO_WGRAD4"] = env
        else: os.environ.pop("FCD_NO_WGRAD4", None)
        for prec in (3, 2):
            d = g.wgrad_desc(n, 2 * h, 2 * h, precision=prec)
            if prec == 3: ops.set_wgrad_amax(d, ga, xa)
            dw = ops.conv_wgrad(d, small, big)
            for _ in range(10): ops.conv_wgrad(d, small, big, dw=dw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): ops.conv_wgrad(d, small, big, dw=dw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            gf = 2.0 * n * h * h * O * I * 16 / 1e9
            print(f"O={O} I={I} n={n} {h}x{h} prec {prec} {L.load().fcd_last_conv_kernel().decode():40s} {us:7.1f} us incl. reduce  {gf / us * 1e-3:6.1f} TF", flush=True)
