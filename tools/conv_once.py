import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import _lib as L, ops
prec = int(sys.argv[1]); sk = int(sys.argv[2])
dev = torch.device('cuda:0')
g = ops.ConvGeom(256, 256, 3, 1, 1)
n, h, w = 16, 32, 32
x = torch.randn(n, h, w, 256, device=dev); wt = torch.randn(256, 256, 3, 3, device=dev) / 48
wf = ops.pack_weight_fwd(wt, g)
d = g.fwd_desc(n, h, w, precision=prec); d.splitk = sk
out = ops.conv_gather(d, x, wf)
for _ in range(5): ops.conv_gather(d, x, wf, out=out)
torch.cuda.synchronize()
