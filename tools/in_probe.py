"""A/B timing of the InstanceNorm kernels within one process (box-to-box HBM variance is ~10-30 %)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
def t(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for shape in ((32, 32, 32, 256), (32, 128, 128, 64)):
    n, h, w, c = shape
    z = torch.randn(shape, device=dev); gy = torch.randn(shape, device=dev)
    gamma = torch.rand(c, device=dev) + 0.5; beta = torch.randn(c, device=dev) * 0.1
    mean, rstd, _ = ops.instnorm_stats(z)
    y = ops.instnorm_apply(z, mean, rstd, gamma, beta, 0.0)
    mb = z.numel() * 4 / 1e6
    for rep in range(1):
        print(shape, "stats      %.1f us" % t(lambda: ops.instnorm_stats(z)), " (%.0f MB)" % mb)
        print(shape, "apply      %.1f us" % t(lambda: ops.instnorm_apply(z, mean, rstd, gamma, beta, 0.0, out=y)))
        print(shape, "bwd with y %.1f us" % t(lambda: ops.instnorm_bwd(gy, y, z, mean, rstd, gamma, 0.0, False)))
        print(shape, "bwd beta   %.1f us" % t(lambda: ops.instnorm_bwd(gy, None, z, mean, rstd, gamma, 0.0, False, beta=beta)))
        print(shape, "bwd slope1 %.1f us" % t(lambda: ops.instnorm_bwd(gy, None, z, mean, rstd, gamma, 1.0, False)), flush=True)
