"""The discriminator's weight-bound deep layers (4x4 stride 2, 512 -> 1024 -> 2048; forward and adjoint forms, 32 and 16
samples): time per launch (gather + split-K reduction) against the split-K factor -- the library's choice (about one
workgroup per CU) and multiples of it.  Are these launches short of loads in flight?   python tools/d_splitk_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
chans = [10, 64, 128, 256, 512, 1024, 2048]
torch.manual_seed(0)


def time_it(fn, n=40):
    for _ in range(8): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n in (32, 16):
    for li in (3, 4, 5):
        I, O, h = chans[li], chans[li + 1], 128 >> li
        g = ops.ConvGeom(O, I, 4, 2, 1)
        wt = torch.randn(O, I, 4, 4, device=dev) / (4 * I ** 0.5)
        for form in ("fwd", "adj"):
            x = torch.randn(n, h, h, g.IP, device=dev) if form == "fwd" else torch.randn(n, h // 2, h // 2, g.OP, device=dev)
            mk = (lambda **k: g.fwd_desc(n, h, h, precision=2, **k)) if form == "fwd" else (lambda **k: g.adj_desc(n, h // 2, h // 2, precision=2, **k))
            d0 = mk()
            fmt = ops.weight_format(d0)
            wf = ((ops.pack_weight_fwd_x3 if form == "fwd" else ops.pack_weight_adj_x3) if fmt == 1 else
                  (ops.pack_weight_fwd if form == "fwd" else ops.pack_weight_adj))(wt, g)
            auto = int(L.load().fcd_conv_resolved_splitk(d0)) if hasattr(L.load(), "fcd_conv_resolved_splitk") else 0
            out = ops.conv_gather(d0, x, wf)
            k0 = L.load().fcd_last_conv_kernel().decode()
            res = []
            for sk in (0, 4, 8, 16, 32, 64):
                d = mk()
                d.splitk = sk
                try:
                    t = time_it(lambda: ops.conv_gather(d, x, wf, out=out))
                except RuntimeError as e:
                    t = float("nan")
                res.append(f"sk{sk}:{t:6.1f}")
            print(f"n={n} L{li+1} {I:4d}->{O:4d} {form} fmt{fmt} {k0[22:60]:40s} " + " ".join(res), flush=True)
