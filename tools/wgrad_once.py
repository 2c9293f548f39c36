"""A few launches of the residual-block weight gradient (3x3 256->256 @32x32, N samples, precision P) for
rocprofv3 --pmc passes:  python tools/wgrad_once.py <prec 0|1|2|3> <N>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
prec = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
g = ops.ConvGeom(256, 256, 3, 1, 1)
x = torch.randn(n, 32, 32, 256, device=dev); gy = torch.randn(n, 32, 32, 256, device=dev)
wd = g.wgrad_desc(n, 32, 32, precision=prec)
if prec == 3: ops.set_wgrad_amax(wd, ops.absmax(gy), ops.absmax(x))
dw = ops.conv_wgrad(wd, gy, x)
for _ in range(5): ops.conv_wgrad(wd, gy, x, dw)
torch.cuda.synchronize()
