"""The discriminator's weight-bound layers (M = 64 .. 1024 output rows, 8 .. 134 MB of weights): time of the forward
convolution and its data gradient per precision mode, with the HBM rate of the weight stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def t_of(fn, iters=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (O, I, n, hs) in [(512, 256, 16, 8), (1024, 512, 16, 4), (2048, 1024, 16, 2), (2048, 1024, 32, 2), (1024, 512, 48, 4)]:
    g = ops.ConvGeom(O, I, 4, 2, 1)
    x = torch.randn(n, 2 * hs, 2 * hs, I, device=dev)
    gy = torch.randn(n, hs, hs, O, device=dev)
    wt = torch.randn(O, I, 4, 4, device=dev) / (4 * I ** 0.5)
    line = f"{(O, I, n, hs)} weights {O * I * 16 * 4 / 1e6:.0f} MB:"
    for prec in (0, 2):
        d = g.fwd_desc(n, 2 * hs, 2 * hs, precision=prec, act_slope=0.01)
        fmt = ops.weight_format(d)
        wp = ops.pack_weight_fwd_x3(wt, g) if fmt == 1 else ops.pack_weight_fwd(wt, g)
        out = ops.conv_gather(d, x, wp)
        tf = t_of(lambda: ops.conv_gather(d, x, wp, out=out))
        kf = L.load().fcd_last_conv_kernel().decode()[22:]
        da = g.adj_desc(n, hs, hs, precision=prec)
        fa = ops.weight_format(da)
        wa = ops.pack_weight_adj_x3(wt, g) if fa == 1 else ops.pack_weight_adj(wt, g)
        gx = ops.conv_gather(da, gy, wa)
        ta = t_of(lambda: ops.conv_gather(da, gy, wa, out=gx))
        ka = L.load().fcd_last_conv_kernel().decode()[22:]
        line += f"  prec {prec}: fwd {tf:.1f} us ({O * I * 16 * (6 if fmt == 1 else 4) / tf / 1e6:.2f} TB/s, fmt {fmt}, sk {d.splitk} {kf}) dgrad {ta:.1f} us ({ka})"
    print(line, flush=True)
