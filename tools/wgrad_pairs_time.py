"""Two-pair weight gradient on the halo kernel (fcd_conv_wgrad2) against two accumulating single-pair launches, on the
generator's 3 x 3 and 4 x 4 layer shapes at the G step's batch (two passes of 32 images): time per layer and the kernel
each form takes.   python tools/wgrad_pairs_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
torch.manual_seed(0)
N = 32
for (k, O, I, h, w) in ((3, 256, 256, 32, 32), (4, 256, 128, 32, 32), (4, 128, 64, 64, 64)):
    g = ops.ConvGeom(O, I, k, 1, 1) if k == 3 else ops.ConvGeom(O, I, 4, 2, 1)
    s = 1 if k == 3 else 2
    xs = [torch.randn(N, s * h, s * w, g.IP, device=dev) for _ in range(2)]
    gs = [torch.randn(N, h, w, g.OP, device=dev) * 0.01 for _ in range(2)]
    mx = [(ops.absmax(a), ops.absmax(b)) for a, b in zip(gs, xs)]
    for prec in (3, 2):
        def desc(j):
            d = g.wgrad_desc(N, s * h, s * w, precision=prec, accumulate=True)
            return ops.set_wgrad_amax(d, *mx[j]) if prec == 3 else d
        dw = torch.zeros(O, I, k, k, device=dev)
        def two():
            ops.conv_wgrad(desc(0), gs[0], xs[0], dw=dw)
            ops.conv_wgrad(desc(1), gs[1], xs[1], dw=dw)
        def one():
            ops.conv_wgrad2(desc(0), gs[0], xs[0], gs[1], xs[1], dw=dw, amax2=mx[1] if prec == 3 else None)
        res = []
        for fn in (two, one):
            for _ in range(5): fn()
            k_ = L.load().fcd_last_conv_kernel().decode()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): fn()
            e1.record(); torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 30 * 1e3, k_))
        print(f"k{k} {I}->{O} {h}x{w} prec {prec}: two launches {res[0][0]:7.1f} us ({res[0][1]}) | one {res[1][0]:7.1f} us ({res[1][1]})", flush=True)
