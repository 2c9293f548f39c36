"""A/B time of the 64-output-channel phase-decomposed adjoint (dgrad of the 4x4 s2 64->128 conv = the
128->64 up-sampling form, 32 samples, 64x64 -> 128x128, fp32x3) between builds of the library."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from fcd_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
g = ops.ConvGeom(128, 64, 4, 2, 1)
n = 32
x = torch.randn(n, 64, 64, 128, device=dev); wt = torch.randn(128, 64, 4, 4, device=dev) / 32
d = g.adj_desc(n, 64, 64, precision=2)
wf = ops.pack_weight_adj_x3(wt, g)
out = ops.conv_gather(d, x, wf)
ref = ops.conv_gather(g.adj_desc(n, 64, 64, precision=0), x, ops.pack_weight_adj(wt, g))
err = float((out - ref).norm() / ref.norm())
for _ in range(100): ops.conv_gather(d, x, wf, out=out)
res = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): ops.conv_gather(d, x, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) * 10, 1))
print(res, 'rel diff to the exact-fp32 kernel', err)
''' % root
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, FCD_HIP_LIB=os.path.join(root, "fcd_amd", lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        print(rnd, lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
