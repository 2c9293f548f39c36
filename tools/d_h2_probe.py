"""The discriminator's six 4x4 stride-2 layers (forward and adjoint form, 32 and 16 samples): today's BF16X3 launch
against the two-plane fp16 form with pre-split weights (FCD_PREC_F16X2; run with FCD_BPRE_MINM=0 so that the
weight-bound layers take pre-split weights too).   python tools/d_h2_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
chans = [10, 64, 128, 256, 512, 1024, 2048]
tot = {2: 0.0, 3: 0.0}
for n in (32, 16):
    for li in range(6):
        I, O, h = chans[li], chans[li + 1], 128 >> li
        g = ops.ConvGeom(O, I, 4, 2, 1)
        wt = torch.randn(O, I, 4, 4, device=dev) / (4 * I ** 0.5)
        for form in ("fwd", "adj"):
            if form == "fwd":
                x = torch.randn(n, h, h, g.IP, device=dev)
            else:
                x = torch.randn(n, h // 2, h // 2, g.OP, device=dev)
            row = []
            for prec in (2, 3):
                d = g.fwd_desc(n, h, h, precision=prec, act_slope=0.01) if form == "fwd" else g.adj_desc(n, h // 2, h // 2, precision=prec)
                fmt = ops.weight_format(d)
                if fmt == 2:
                    wa = ops.absmax(wt); ops.set_amax(d, ops.absmax(x), wa)
                    wf = (ops.pack_weight_fwd_h2 if form == "fwd" else ops.pack_weight_adj_h2)(wt, g, wa)
                elif fmt == 1:
                    wf = (ops.pack_weight_fwd_x3 if form == "fwd" else ops.pack_weight_adj_x3)(wt, g)
                else:
                    wf = (ops.pack_weight_fwd if form == "fwd" else ops.pack_weight_adj)(wt, g)
                out = ops.conv_gather(d, x, wf)
                for _ in range(20): ops.conv_gather(d, x, wf, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(100): ops.conv_gather(d, x, wf, out=out)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 100 * 1e3
                tot[prec] += us
                row.append(f"prec {prec} fmt {fmt} {L.load().fcd_last_conv_kernel().decode()[22:]:42s} {us:6.1f} us")
            print(f"n={n} L{li+1} {I:4d}->{O:4d} {form}: " + " | ".join(row), flush=True)
print("sum over these launches: BF16X3 %.0f us, F16X2 %.0f us" % (tot[2], tot[3]))
