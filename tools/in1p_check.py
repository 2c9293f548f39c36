"""One-pass InstanceNorm backward (32 x 32 images) against the three-kernel form (FCD_NO_IN1P=1): dz, dgamma, dbeta, the
fused max |dz|, and time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')


def run(n, c, slope, use_running, onepass, iters=0):
    if onepass: os.environ.pop("FCD_NO_IN1P", None)
    else: os.environ["FCD_NO_IN1P"] = "1"
    gen = torch.Generator(device=dev).manual_seed(2)
    z = torch.randn(n, 32, 32, c, device=dev, generator=gen) * 2 + 0.3
    gy = torch.randn(n, 32, 32, c, device=dev, generator=gen) * 1e-3
    gamma = torch.rand(c, device=dev, generator=gen) + 0.5
    beta = torch.randn(c, device=dev, generator=gen) * 0.3
    mean, rstd, _ = ops.instnorm_stats(z)
    am = torch.zeros(ops.AMAX_SLOTS, device=dev)
    dz, dg, db = ops.instnorm_bwd(gy, None, z, mean, rstd, gamma, slope, use_running, beta=beta, dz_amax=am)
    t = None
    if iters:
        for _ in range(20): ops.instnorm_bwd(gy, None, z, mean, rstd, gamma, slope, use_running, beta=beta)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.instnorm_bwd(gy, None, z, mean, rstd, gamma, slope, use_running, beta=beta)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
    return dz, dg, db, am, t


for (n, c, slope, ur) in [(32, 256, 0.0, False), (16, 256, 0.0, False), (32, 256, 1.0, False), (4, 512, 0.01, False), (32, 256, 0.0, True), (8, 64, 0.0, False)]:
    a = run(n, c, slope, ur, False, 200)
    b = run(n, c, slope, ur, True, 200)
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / y.double().abs().max())
    print((n, c, slope, ur), f"dz {rel(b[0], a[0]):.2e} dgamma {rel(b[1], a[1]):.2e} dbeta {rel(b[2], a[2]):.2e} amax {float(a[3].max()):.6e} / {float(b[3].max()):.6e}"
          f" (= max|dz| {float(b[0].abs().max()):.6e})  three kernels {a[4]:.1f} us  one pass {b[4]:.1f} us", flush=True)
