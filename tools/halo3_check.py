"""A-halo form of the 256x128 BF16X3 gather kernel (3x3, stride 1, pad 1; activations split once per 32-channel slice
into bf16 planes of the tile's halo region) against the per-tap gather form (FCD_NO_HALO3=1): bits and time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')


def run(O, I, n, h, w, form, halo, iters=0):
    if halo: os.environ.pop("FCD_NO_HALO3", None)
    else: os.environ["FCD_NO_HALO3"] = "1"
    g = ops.ConvGeom(O, I, 3, 1, 1)
    gen = torch.Generator(device=dev).manual_seed(1)
    cin = I if form == "fwd" else O
    x = torch.randn(n, h, w, cin, device=dev, generator=gen)
    wt = torch.randn(O, I, 3, 3, device=dev, generator=gen) / (3 * I ** 0.5)
    mask = torch.randn(n, h, w, I if form == "adj" else O, device=dev, generator=gen)
    if form == "fwd":
        d = g.fwd_desc(n, h, w, precision=2)
        wp = ops.pack_weight_fwd_x3(wt, g)
        out, mean, rstd, _ = ops.conv_gather_stats(d, x, wp, want_var=True)
    else:
        d = g.adj_desc(n, h, w, precision=2, mask_slope=0.0)
        wp = ops.pack_weight_adj_x3(wt, g)
        out = ops.conv_gather(d, x, wp, mask=mask, residual=mask)
        mean = rstd = out[0, 0, 0]
    name = L.load().fcd_last_conv_kernel().decode()
    t = None
    if iters:
        for _ in range(50): ops.conv_gather(d, x, wp, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): ops.conv_gather(d, x, wp, out=out)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
        if form == "fwd": out, mean, rstd, _ = ops.conv_gather_stats(d, x, wp, want_var=True)
        else: out = ops.conv_gather(d, x, wp, mask=mask, residual=mask)
    return out, mean, rstd, name, t


for (O, I, n, h, w, form) in [(256, 256, 32, 32, 32, "fwd"), (256, 256, 32, 32, 32, "adj"), (128, 64, 8, 64, 64, "fwd"), (128, 128, 9, 64, 64, "adj"),
                              (256, 96, 40, 16, 32, "fwd"), (512, 512, 8, 64, 64, "fwd"), (256, 256, 64, 32, 32, "fwd"),
                              (256, 256, 16, 32, 32, "fwd"), (256, 256, 16, 32, 32, "adj"), (128, 96, 6, 12, 32, "fwd"), (256, 64, 5, 30, 64, "adj")]:
    a, ma, ra, na, ta = run(O, I, n, h, w, form, False, 100)
    b, mb, rb, nb, tb = run(O, I, n, h, w, form, True, 100)
    print((O, I, n, h, w, form), na[22:], "->", nb[22:], "same bits", torch.equal(a, b), torch.equal(ma, mb), torch.equal(ra, rb),
          "max diff", float((a - b).abs().max()), f"gather {ta:.1f} us  halo {tb:.1f} us", flush=True)
