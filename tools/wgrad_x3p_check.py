"""Split-once weight gradient (conv_wgrad_x3p_kernel) against the in-register-split kernel (FCD_WGRAD_OLD=1):
bits and time, on the residual-block shape and a few ragged ones."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)


def run(shape, old, iters=0):
    O, I, k, s, pd, n, h, w = shape
    if old:
        os.environ["FCD_WGRAD_OLD"] = "1"
    else:
        os.environ.pop("FCD_WGRAD_OLD", None)
    g = ops.ConvGeom(O, I, k, s, pd)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, g.IP, generator=gen).to(dev)
    sh, sw = g.small_hw(h, w)
    gy = torch.randn(n, sh, sw, g.OP, generator=gen).to(dev)
    wd = g.wgrad_desc(n, h, w, precision=2)
    dw = ops.conv_wgrad(wd, gy, x)
    t = None
    if iters:
        for _ in range(20): ops.conv_wgrad(wd, gy, x, dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.conv_wgrad(wd, gy, x, dw)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
    return dw, t


for shape in [(256, 256, 3, 1, 1, 32, 32, 32), (256, 256, 3, 1, 1, 16, 32, 32), (256, 128, 4, 2, 1, 32, 64, 64),
              (300, 36, 3, 1, 1, 3, 17, 13), (128, 64, 4, 2, 1, 32, 128, 128), (128, 256, 4, 2, 1, 32, 64, 64), (64, 128, 4, 2, 1, 32, 128, 128), (64, 12, 7, 1, 3, 32, 128, 128), (100, 20, 3, 1, 1, 2, 9, 11), (40, 8, 4, 2, 1, 5, 10, 6), (512, 256, 4, 2, 1, 16, 16, 16), (2048, 1024, 4, 2, 1, 16, 4, 4)]:
    a, ta = run(shape, True, 50)
    b, tb = run(shape, False, 50)
    ref = None
    print(shape, "identical bits:", torch.equal(a, b), "max diff", float((a - b).abs().max()), "rel", float((a - b).norm() / a.norm()),
          f"old {ta:.1f} us  new {tb:.1f} us", flush=True)
