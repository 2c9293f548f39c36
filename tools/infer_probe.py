"""The inference leg alone (G eval + difference map + threshold on a resident batch of 64) for a kernel trace:
python3 tools/infer_probe.py [precision] [iters]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32x2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
s = FCDSolver(None, c_dim=1, image_size=128, g_conv_dim=64, d_conv_dim=64, g_repeat_num=6, d_repeat_num=6, num_channels=10,
              batch_size=16)
s.to(dev)
s.set_precision(prec)
x = synth_batch(64, 128, 10, seed=7)[0][0].to(dev)
s.G.eval()
with torch.no_grad():
    for _ in range(3): s.difference_and_mask(x, 0.027)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): s.difference_and_mask(x, 0.027)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
print(f"{prec}: {dt * 1e3:.2f} ms per batch of 64, {64 / dt:.0f} images/s", flush=True)
