"""A/B time of the residual-block weight gradient (3x3 256->256 @32x32, AB_N samples (default 32), fp32x3,
split-K slabs + reduction included) between builds of the library (FCD_HIP_LIB per child process)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from fcd_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
g = ops.ConvGeom(256, 256, 3, 1, 1)
n, h, w = int(os.environ.get("AB_N", "32")), 32, 32
x = torch.randn(n, h, w, 256, device=dev); gy = torch.randn(n, h, w, 256, device=dev)
wd = g.wgrad_desc(n, h, w, precision=2)
dw = ops.conv_wgrad(wd, gy, x)
ref = ops.conv_wgrad(g.wgrad_desc(n, h, w, precision=0), gy, x)
err = float((dw - ref).norm() / ref.norm())
import hashlib
sig = hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()[:12]
for _ in range(100): ops.conv_wgrad(wd, gy, x, dw)
res = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): ops.conv_wgrad(wd, gy, x, dw)
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) * 10, 1))
print(res, 'rel diff to the exact-fp32 kernel', err, 'sha', sig)
''' % root
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, FCD_HIP_LIB=os.path.join(root, "fcd_amd", lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        print(rnd, lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
