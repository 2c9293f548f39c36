import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/fcd_amd") else os.getcwd())
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
def t_of(fn, iters=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (O, I, n, hs) in [(512, 256, 16, 8), (1024, 512, 16, 4), (2048, 1024, 16, 2)]:
    g = ops.ConvGeom(O, I, 4, 2, 1)
    x = torch.randn(n, 2 * hs, 2 * hs, I, device=dev)
    wt = torch.randn(O, I, 4, 4, device=dev) / (4 * I ** 0.5)
    wp = ops.pack_weight_fwd(wt, g)
    line = f"{(O, I, n, hs)}:"
    for sk in (0, 2, 4, 8, 16, 32, 64):
        d = g.fwd_desc(n, 2 * hs, 2 * hs, precision=2, act_slope=0.01)
        d.splitk = sk
        try:
            out = ops.conv_gather(d, x, wp)
            line += f"  sk{sk} {t_of(lambda: ops.conv_gather(d, x, wp, out=out)):.1f}"
        except Exception as e:
            line += f"  sk{sk} err"
    print(line, flush=True)
