"""Time of the four-phase 4x4 stride-2 adjoint (ConvTranspose2d / data gradient of the strided convolutions) at the
generator's two up-sampling shapes, 32 samples, in one precision with the library FCD_HIP_LIB names.
   python tools/adj_time.py [precision code, default 3]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
for (O, I, n, h) in [(128, 64, 32, 64), (256, 128, 32, 32), (128, 64, 16, 64), (256, 128, 16, 32)]:
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gy = torch.randn(n, h, h, O, device=dev); wt = torch.randn(O, I, 4, 4, device=dev) / 32
    d = g.adj_desc(n, h, h, precision=prec)
    fmt = ops.weight_format(d)
    if fmt == 2:
        wa = ops.absmax(wt); ops.set_amax(d, ops.absmax(gy), wa); wf = ops.pack_weight_adj_h2(wt, g, wa)
    else:
        wf = ops.pack_weight_adj_x3(wt, g) if fmt == 1 else ops.pack_weight_adj(wt, g)
    out = ops.conv_gather(d, gy, wf)
    for _ in range(100): ops.conv_gather(d, gy, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(300): ops.conv_gather(d, gy, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("FCD_HIP_LIB", "default"), (O, I, n, h), L.load().fcd_last_conv_kernel().decode()[22:], f"{e0.elapsed_time(e1) / 300 * 1e3:.1f} us", flush=True)
