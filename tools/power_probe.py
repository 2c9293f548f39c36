"""Same resblock conv launch on random, small-magnitude and all-zero operands: separates a
software stall from a data-dependent (power) limit on the fp32 MFMA rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import _lib as L, ops
dev = torch.device('cuda:0')
g = ops.ConvGeom(256, 256, 3, 1, 1)
n, h, w = 16, 32, 32
for mode in ('random', 'zeros', 'ones', 'random', 'sparse'):
    if mode == 'random':
        x = torch.randn(n, h, w, 256, device=dev); wt = torch.randn(256, 256, 3, 3, device=dev) / 48
    elif mode == 'zeros':
        x = torch.zeros(n, h, w, 256, device=dev); wt = torch.zeros(256, 256, 3, 3, device=dev)
    elif mode == 'ones':
        x = torch.ones(n, h, w, 256, device=dev); wt = torch.ones(256, 256, 3, 3, device=dev)
    else:
        x = torch.randn(n, h, w, 256, device=dev) * (torch.rand(n, h, w, 256, device=dev) < 0.1)
        wt = torch.randn(256, 256, 3, 3, device=dev) / 48
    wf = ops.pack_weight_fwd(wt, g)
    d = g.fwd_desc(n, h, w); d.splitk = 1
    out = ops.conv_gather(d, x, wf)
    for _ in range(10): ops.conv_gather(d, x, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): ops.conv_gather(d, x, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(mode, 'us', round(us, 1), 'TF', round(19.327e3 / us, 1), flush=True)
