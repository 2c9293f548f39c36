"""Register-staged gather kernel (template flag RS of conv_gather_dma_kernel) on the discriminator's 4x4 stride-2 layers,
forward and adjoint form: bits against the in-register-split form (FCD_NO_RS=1), error of the scaled-fp16 instances
against float64-accumulated torch, and time per launch of every variant.   python tools/rs_check.py [time]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
chans = [10, 64, 128, 256, 512, 1024, 2048]
timing = len(sys.argv) > 1


def run(d, x, wt, g, form, out=None, bias=None, mask=None):
    fmt = ops.weight_format(d)
    if fmt in (2, 3):
        wa = ops.absmax(wt)
        ops.set_amax(d, ops.absmax(x), wa)
    if fmt == 2:
        wf = (ops.pack_weight_fwd_h2 if form == "fwd" else ops.pack_weight_adj_h2)(wt, g, wa)
    elif fmt == 1:
        wf = (ops.pack_weight_fwd_x3 if form == "fwd" else ops.pack_weight_adj_x3)(wt, g)
    else:
        wf = (ops.pack_weight_fwd if form == "fwd" else ops.pack_weight_adj)(wt, g)
    return ops.conv_gather(d, x, wf, out=out, bias=bias, mask=mask), wf, fmt


def time_it(d, x, wf, out):
    for _ in range(10): ops.conv_gather(d, x, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): ops.conv_gather(d, x, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


torch.manual_seed(0)
worst = 0.0
for n in (32, 16):
    for li in range(1, 6):
        I, O, h = chans[li], chans[li + 1], 128 >> li
        g = ops.ConvGeom(O, I, 4, 2, 1)
        wt = torch.randn(O, I, 4, 4, device=dev) / (4 * I ** 0.5)
        for form in ("fwd", "adj"):
            x = torch.randn(n, h, h, g.IP, device=dev) if form == "fwd" else torch.randn(n, h // 2, h // 2, g.OP, device=dev)
            x = torch.where(x > 0, x, 0.01 * x)
            mk = (lambda p, **k: g.fwd_desc(n, h, h, precision=p, **k)) if form == "fwd" else (lambda p, **k: g.adj_desc(n, h // 2, h // 2, precision=p, **k))
            bias = torch.randn(O if form == "fwd" else I, device=dev) if form == "fwd" else None
            os.environ["FCD_NO_RS"] = "1"
            ref, wf0, fmt0 = run(mk(2, act_slope=0.01), x, wt, g, form, bias=bias)
            k0 = L.load().fcd_last_conv_kernel().decode()
            t0 = time_it(mk(2, act_slope=0.01), x, wf0, ref.clone()) if timing else 0.0
            del os.environ["FCD_NO_RS"]
            got, wf1, fmt1 = run(mk(2, act_slope=0.01), x, wt, g, form, bias=bias)
            k1 = L.load().fcd_last_conv_kernel().decode()
            same = bool(torch.equal(ref, got))
            t1 = time_it(mk(2, act_slope=0.01), x, wf1, got.clone()) if timing else 0.0
            d3 = mk(3, act_slope=0.01)
            h2, wf3, fmt3 = run(d3, x, wt, g, form, bias=bias)
            k3 = L.load().fcd_last_conv_kernel().decode()
            t3 = time_it(d3, x, wf3, h2.clone()) if timing else 0.0
            err = float((h2.double() - ref.double()).norm() / ref.double().norm())
            worst = max(worst, err)
            rs = "true>" in k1
            print(f"n={n} L{li+1} {I:4d}->{O:4d} {form}: x3 {'RS ' if rs else '-- '} bits {'same' if same else 'DIFFER'} "
                  f"{t0:6.1f} -> {t1:6.1f} us | h2 fmt {fmt3} {k3[22:]:44s} {t3:6.1f} us  rel {err:.2e}", flush=True)
            if rs and not same:
                print("   max abs diff", float((ref - got).abs().max()))
print("worst h2-vs-x3 relative difference %.2e" % worst)
