"""A/B of the discriminator's (4x4, stride 2) and the up / down-sampling weight gradients between builds of the library
(FCD_HIP_LIB per child process, builds alternate): python3 tools/ab_wgrad_d.py libfcd_prev.so libfcd_hip.so"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch, hashlib
sys.path.insert(0, %r)
from fcd_amd import ops, _lib as L
dev = torch.device('cuda:0')
for (O, I, k, s, p, n, h, w) in [(64, 12, 4, 2, 1, 32, 128, 128), (128, 64, 4, 2, 1, 32, 64, 64), (256, 128, 4, 2, 1, 32, 32, 32),
                                 (512, 256, 4, 2, 1, 32, 16, 16), (1024, 512, 4, 2, 1, 32, 8, 8), (128, 64, 4, 2, 1, 16, 128, 128)]:
    g = ops.ConvGeom(O, I, k, s, p)
    gen = torch.Generator(device=dev).manual_seed(1)
    sh, sw = g.small_hw(h, w)
    x = torch.randn(n, h, w, g.IP, device=dev, generator=gen); gy = torch.randn(n, sh, sw, g.OP, device=dev, generator=gen)
    wd = g.wgrad_desc(n, h, w, precision=2)
    dw = ops.conv_wgrad(wd, gy, x)
    name = L.load().fcd_last_conv_kernel().decode()
    sig = hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()[:10]
    for _ in range(30): ops.conv_wgrad(wd, gy, x, dw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): ops.conv_wgrad(wd, gy, x, dw)
    e1.record(); torch.cuda.synchronize()
    print((O, I, n, h), name, "%%.1f us" %% (e0.elapsed_time(e1) * 10), sig, flush=True)
''' % root
for lib in sys.argv[1:]:
    env = dict(os.environ, FCD_HIP_LIB=os.path.join(root, "fcd_amd", lib))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(lib); print(out.stdout if out.stdout.strip() else out.stderr[-400:], flush=True)
