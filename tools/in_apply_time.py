"""in_apply_kernel alone at the step's shapes: time per launch (with and without residual / fused maxima) and a
checksum of the output, for A/B between builds (FCD_HIP_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
for (n, hw, c, res, amax) in [(32, 32, 256, False, True), (32, 32, 256, True, True), (16, 32, 256, False, True),
                              (32, 64, 128, False, True), (32, 128, 64, False, True), (32, 32, 256, False, False),
                              (3, 17, 20, True, True)]:
    gen = torch.Generator(device=dev).manual_seed(4)
    z = torch.randn(n, hw, hw, c, device=dev, generator=gen) * 2 + 0.3
    r = torch.randn(n, hw, hw, c, device=dev, generator=gen) if res else None
    gamma = torch.rand(c, device=dev, generator=gen) + 0.5
    beta = torch.randn(c, device=dev, generator=gen) * 0.3
    mean, rstd, _ = ops.instnorm_stats(z)
    out = torch.empty_like(z)
    am = torch.zeros(ops.AMAX_SLOTS, device=dev) if amax else None
    kw = dict(out=out, residual=r)
    if amax: kw["amax"] = am
    for _ in range(20): ops.instnorm_apply(z, mean, rstd, gamma, beta, 0.0, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): ops.instnorm_apply(z, mean, rstd, gamma, beta, 0.0, **kw)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 200 * 1e3
    mb = z.numel() * 4 * (3 if res else 2) / 1e6
    ref = torch.relu((z - mean[:, None, None, :]) * rstd[:, None, None, :] * gamma + beta) + (r if res else 0)
    print((n, hw, c, res, amax), f"{t:.1f} us  {mb / t:.2f} TB/s  sum {float(out.double().sum()):.10e} max err {float((out - ref).abs().max()):.2e}"
          + (f" amax {float(am.max()):.6e} = {float(out.abs().max()):.6e}" if amax else ""), flush=True)
