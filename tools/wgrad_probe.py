import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import _lib as L, ops
dev = torch.device('cuda:0')
def run(O, I, k, s, p, n, h, w, sks):
    g = ops.ConvGeom(O, I, k, s, p)
    sh, sw = g.small_hw(h, w)
    x = torch.randn(n, h, w, g.IP, device=dev); gy = torch.randn(n, sh, sw, g.OP, device=dev)
    flops = 2.0 * n * sh * sw * O * I * k * k
    for sk in sks:
        d = g.wgrad_desc(n, h, w); d.splitk = sk
        dw = ops.conv_wgrad(d, gy, x)
        for _ in range(3): ops.conv_wgrad(d, gy, x, dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): ops.conv_wgrad(d, gy, x, dw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"wgrad O{O} I{I} k{k} s{s} {h}x{w} splitk {sk}: {us:.1f} us (incl. reduce)  {flops / us / 1e6:.1f} TF", flush=True)
run(10, 64, 7, 1, 3, 16, 128, 128, (0, 10, 20, 30, 40, 60, 80))
run(64, 12, 7, 1, 3, 16, 128, 128, (0, 26, 51, 102))
run(256, 256, 3, 1, 1, 16, 32, 32, (0, 14))
