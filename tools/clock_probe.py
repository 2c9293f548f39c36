"""Engine clock while the residual-block convolution (32 samples) runs back to back for a few seconds: a
child process loops the kernel, this process polls rocm-smi (read-only).
    python tools/clock_probe.py <precision 0|1|2|3> [wgrad]"""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prec = sys.argv[1] if len(sys.argv) > 1 else "2"
wgrad = len(sys.argv) > 2 and sys.argv[2] == "wgrad"
code = r'''
import sys, time, torch
sys.path.insert(0, %r)
from fcd_amd import ops
dev = torch.device('cuda:0')
g = ops.ConvGeom(256, 256, 3, 1, 1)
x = torch.randn(32, 32, 32, 256, device=dev); wt = torch.randn(256, 256, 3, 3, device=dev) / 48
prec = %s
d = g.fwd_desc(32, 32, 32, precision=prec)
fmt = ops.weight_format(d)
if fmt == 2:
    wa = ops.absmax(wt)
    ops.set_amax(d, ops.absmax(x), wa)
    wf = ops.pack_weight_fwd_h2(wt, g, wa)
else:
    wf = ops.pack_weight_fwd_x3(wt, g) if fmt == 1 else ops.pack_weight_fwd(wt, g)
if %s:
    gy = torch.randn(32, 32, 32, 256, device=dev)
    wd = g.wgrad_desc(32, 32, 32, precision=prec)
    if prec == 3: ops.set_wgrad_amax(wd, ops.absmax(gy), ops.absmax(x))
    dw = ops.conv_wgrad(wd, gy, x)
    run = lambda: ops.conv_wgrad(wd, gy, x, dw)
else:
    out = ops.conv_gather(d, x, wf)
    run = lambda: ops.conv_gather(d, x, wf, out=out)
torch.cuda.synchronize(); print("go", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    for _ in range(500): run()
    torch.cuda.synchronize(); n += 500
print("us per launch", (time.time() - t0) / n * 1e6, flush=True)
''' % (root, prec, wgrad)
child = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
print(child.stdout.readline().strip(), flush=True)
for i in range(7):
    time.sleep(1)
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=30)
    keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "fclk"))]
    print(i, " | ".join(keep), flush=True)
    if i == 3:
        print(subprocess.run(["rocm-smi"], capture_output=True, text=True, timeout=30).stdout, flush=True)
print(child.stdout.read().strip(), flush=True)
child.wait(timeout=60)
