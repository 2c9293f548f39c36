"""A/B time of the generator's last convolution (7x7, 64 -> 10, 128x128, AB_N samples, fp32x3 descriptor)
between builds of the library (FCD_HIP_LIB per child process)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from fcd_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
n = int(os.environ.get("AB_N", "32"))
g = ops.ConvGeom(10, 64, 7, 1, 3)
x = torch.randn(n, 128, 128, 64, device=dev); wt = torch.randn(10, 64, 7, 7, device=dev) / 56
d = g.fwd_desc(n, 128, 128, precision=2)
wf = ops.pack_weight_fwd(wt, g)
out = ops.conv_gather(d, x, wf)
for _ in range(30): ops.conv_gather(d, x, wf, out=out)
res = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(40): ops.conv_gather(d, x, wf, out=out)
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) * 25, 1))
print(res, 'checksum', float(out.double().sum()))
''' % root
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, FCD_HIP_LIB=os.path.join(root, "fcd_amd", lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        print(rnd, lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
