import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import _lib as L, ops
prec = int(sys.argv[1]); sk = int(sys.argv[2])
dev = torch.device('cuda:0')
g = ops.ConvGeom(256, 256, 3, 1, 1)
n, h, w = (int(sys.argv[3]) if len(sys.argv) > 3 else 16), 32, 32
x = torch.randn(n, h, w, 256, device=dev); wt = torch.randn(256, 256, 3, 3, device=dev) / 48
d = g.fwd_desc(n, h, w, precision=prec); d.splitk = sk
fmt = ops.weight_format(d)
if fmt == 2:
    wa = ops.absmax(wt)
    ops.set_amax(d, ops.absmax(x), wa)
    wf = ops.pack_weight_fwd_h2(wt, g, wa)
else:
    wf = ops.pack_weight_fwd_x3(wt, g) if fmt == 1 else ops.pack_weight_fwd(wt, g)
out = ops.conv_gather(d, x, wf)
for _ in range(5): ops.conv_gather(d, x, wf, out=out)
torch.cuda.synchronize()
