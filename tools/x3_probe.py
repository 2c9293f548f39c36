"""Accuracy (vs an fp64 convolution) and speed of the three MFMA precision modes on the residual-block conv."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
n, h, w = 16, 32, 32
g = ops.ConvGeom(256, 256, 3, 1, 1)
x = torch.randn(n, 256, h, w); wt = torch.randn(256, 256, 3, 3) / 48
ref = F.conv2d(x.double(), wt.double(), None, 1, 1).permute(0, 2, 3, 1)
xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
wf32 = ops.pack_weight_fwd(wt.to(dev), g)
gy = torch.randn(n, h, w, 256, device=dev)
wref = None
for prec, name in ((0, 'fp32 MFMA'), (2, 'bf16x3'), (1, 'bf16')):
    d = g.fwd_desc(n, h, w, precision=prec)
    wf = ops.pack_weight_fwd_x3(wt.to(dev), g) if ops.wants_split_weights(d) else wf32
    out = ops.conv_gather(d, xd, wf)
    err = float((out.cpu().double() - ref).norm() / ref.norm())
    mx = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    for _ in range(5): ops.conv_gather(d, xd, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): ops.conv_gather(d, xd, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    wd = g.wgrad_desc(n, h, w, precision=prec)
    dw = ops.conv_wgrad(wd, gy, xd)
    if wref is None:
        t = torch.zeros(256, 256, 3, 3, dtype=torch.float64, requires_grad=True)
        (F.conv2d(x.double(), t, None, 1, 1) * gy.cpu().double().permute(0, 3, 1, 2)).sum().backward()
        wref = t.grad
    werr = float((dw.cpu().double() - wref).norm() / wref.norm())
    for _ in range(3): ops.conv_wgrad(wd, gy, xd, dw)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.conv_wgrad(wd, gy, xd, dw)
    e1.record(); torch.cuda.synchronize()
    wus = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:10s} conv rel-err {err:.2e} (max {mx:.2e}) {us:7.1f} us {19.327e3 / us:6.1f} TF | wgrad rel-err {werr:.2e} {wus:7.1f} us", flush=True)
