import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import _lib as L, ops
dev = torch.device('cuda:0')
n, h, w = 16, 32, 32
for k, pad in ((1, 0), (3, 1), (5, 2), (7, 3)):
    g = ops.ConvGeom(256, 256, k, 1, pad)
    x = torch.randn(n, h, w, 256, device=dev); wt = torch.randn(256, 256, k, k, device=dev) / 48
    wf = ops.pack_weight_fwd(wt, g)
    d = g.fwd_desc(n, h, w); d.splitk = 1
    out = ops.conv_gather(d, x, wf)
    for _ in range(3): ops.conv_gather(d, x, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.conv_gather(d, x, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('k', k, 'ktiles', k * k * 8, 'us', round(us, 1), 'us/ktile', round(us / (k * k * 8), 3), flush=True)
