"""Step time of BASELINE.json configs[4] (image 256, batch 32, conv_dim 128: 4465 GFLOP/img/step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32x3'
torch.manual_seed(5)
s = FCDSolver(None, image_size=256, g_conv_dim=128, d_conv_dim=128, batch_size=32, precision=prec).to(dev)
batch, _ = synth_batch(32, 256, 10, seed=3); b = tuple(t.to(dev) for t in batch)
for _ in range(2): s.train_step(b, force_g=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 4
for _ in range(n): s.train_step(b, force_g=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"stress {prec}: {dt * 1e3:.1f} ms/step, {32 / dt:.1f} images/s, {4465e9 * 32 / dt / 1e12:.1f} TFLOP/s, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
