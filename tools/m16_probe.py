"""Split-K sweep of the 10-channel weight gradient (G's last conv: dW[10][64][7][7], 128x128, N samples)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = ops.ConvGeom(10, 64, 7, 1, 3)
x = torch.randn(n, 128, 128, 64, device=dev); gy = torch.randn(n, 128, 128, g.OP, device=dev)
for sk in (0, 8, 16, 32, 64, 128):
    wd = g.wgrad_desc(n, 128, 128, precision=prec); wd.splitk = sk
    dw = ops.conv_wgrad(wd, gy, x)
    for _ in range(3): ops.conv_wgrad(wd, gy, x, dw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): ops.conv_wgrad(wd, gy, x, dw)
    e1.record(); torch.cuda.synchronize()
    print(f"N={n} prec={prec} splitk={sk}: {e0.elapsed_time(e1) * 100:.1f} us", flush=True)
