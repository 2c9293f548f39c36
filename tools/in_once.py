"""A few launches of in_apply_kernel on the (16,128,128,64) activation, for rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE: HBM-side bytes of the bandwidth-bound kernel bench.py reports as roofline_hbm)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
z = torch.randn(16, 128, 128, 64, device=dev)
mean, rstd, _ = ops.instnorm_stats(z)
gamma, beta = torch.ones(64, device=dev), torch.zeros(64, device=dev)
out = torch.empty_like(z)
for _ in range(5): ops.instnorm_apply(z, mean, rstd, gamma, beta, 0.0, out=out)
torch.cuda.synchronize()
