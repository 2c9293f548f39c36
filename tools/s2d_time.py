"""The generator's two 4 x 4 stride-2 down-sampling convolutions (64 -> 128 at 128^2 -> 64^2, 128 -> 256 at 64^2 -> 32^2) at 32
and 64 samples: the space-to-depth A-halo form against the per-tap kernel (FCD_NO_S2D=1).   python tools/s2d_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
for (O, I, hs) in ((128, 64, 64), (256, 128, 32)):
    g = ops.ConvGeom(O, I, 4, 2, 1)
    wd = torch.randn(O, I, 4, 4, device=dev) / (4 * I ** 0.5)
    wa = ops.absmax(wd)
    for n in (32, 64):
        x = torch.randn(n, 2 * hs, 2 * hs, g.IP, device=dev)
        xa = ops.absmax(x)
        for env in ("1", None):
            if env: os.environ["FCD_NO_S2D"] = env
            else: os.environ.pop("FCD_NO_S2D", None)
            for prec in (3, 2):
                d = g.fwd_desc(n, 2 * hs, 2 * hs, precision=prec)
                if prec == 3:
                    ops.set_amax(d, xa, wa); wp = ops.pack_weight_fwd_h2(wd, g, wa)
                else:
                    wp = ops.pack_weight_fwd_x3(wd, g)
                out = ops.conv_gather(d, x, wp)
                for _ in range(10): ops.conv_gather(d, x, wp, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50): ops.conv_gather(d, x, wp, out=out)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 50 * 1e3
                gf = 2.0 * n * hs * hs * O * I * 16 / 1e9
                print(f"{I}->{O} n={n} prec {prec} {L.load().fcd_last_conv_kernel().decode()[22:]:66s} {us:7.1f} us  {gf / us * 1e3:6.1f} TF", flush=True)
