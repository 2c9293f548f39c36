"""Time of the 256->256 3x3 residual-block convolution (32 samples) in one precision with the library FCD_HIP_LIB names
(diagnostic builds: results are not checked).   python tools/h2_time.py [precision code, default 3]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
g = ops.ConvGeom(256, 256, 3, 1, 1)
NS = int(os.environ.get("H2_N", "32"))
x = torch.randn(NS, 32, 32, 256, device=dev); wt = torch.randn(256, 256, 3, 3, device=dev) / 48
d = g.fwd_desc(NS, 32, 32, precision=prec)
fmt = ops.weight_format(d)
if fmt == 2:
    wa = ops.absmax(wt); ops.set_amax(d, ops.absmax(x), wa); wf = ops.pack_weight_fwd_h2(wt, g, wa)
else:
    wf = ops.pack_weight_fwd_x3(wt, g) if fmt == 1 else ops.pack_weight_fwd(wt, g)
out = ops.conv_gather(d, x, wf)
for _ in range(300): ops.conv_gather(d, x, wf, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(500): ops.conv_gather(d, x, wf, out=out)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("FCD_HIP_LIB", "default"), L.load().fcd_last_conv_kernel().decode()[22:], f"{e0.elapsed_time(e1) / 500 * 1e3:.1f} us", flush=True)
