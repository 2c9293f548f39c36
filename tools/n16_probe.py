"""Time of the generator's last convolution (7x7, 64 -> 10, 128x128, N samples) in the given precision."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = ops.ConvGeom(10, 64, 7, 1, 3)
x = torch.randn(n, 128, 128, 64, device=dev); wt = torch.randn(10, 64, 7, 7, device=dev) / 56
for prec in (0, 2):
    d = g.fwd_desc(n, 128, 128, precision=prec)
    wf = ops.pack_weight_fwd_x3(wt, g) if ops.wants_split_weights(d) else ops.pack_weight_fwd(wt, g)
    out = ops.conv_gather(d, x, wf)
    for _ in range(5): ops.conv_gather(d, x, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.conv_gather(d, x, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"N={n} prec={prec}: {e0.elapsed_time(e1) * 50:.1f} us", flush=True)
