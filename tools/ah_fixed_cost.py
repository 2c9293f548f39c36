"""Fixed cost of the dominant A-halo launch: the 3x3 convolution to 256 channels on 32 samples of 32 x 32 with 32 .. 512
input channels (1 .. 16 slices of the same body) -- the intercept of time against slices is what prologue, epilogue
(output + InstanceNorm sums) and launch cost, the slope what a slice (nine taps) costs.   python tools/ah_fixed_cost.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
L.check(L.load().fcd_init(0), "init")
res = []
for cin in (32, 64, 128, 256, 512):
    g = ops.ConvGeom(256, cin, 3, 1, 1)
    x = torch.randn(32, 32, 32, cin, device=dev); wt = torch.randn(256, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    d = g.fwd_desc(32, 32, 32, precision=3)
    wa = ops.absmax(wt); ops.set_amax(d, ops.absmax(x), wa); wf = ops.pack_weight_fwd_h2(wt, g, wa)
    out = ops.conv_gather(d, x, wf)
    k = L.load().fcd_last_conv_kernel().decode()[22:]
    for _ in range(100): ops.conv_gather(d, x, wf, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(300): ops.conv_gather(d, x, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 300 * 1e3
    res.append((cin // 32, t))
    print(f"Cin {cin:4d} ({cin // 32:2d} slices) {k}: {t:6.1f} us", flush=True)
(s0, t0), (s1, t1) = res[0], res[-1]
slope = (t1 - t0) / (s1 - s0)
print(f"per slice {slope:.2f} us, intercept {t0 - slope * s0:.1f} us; the 8-slice launch: {8 * slope:.1f} us of slices + intercept")
