"""Step time of BASELINE.json configs[4] (image 256, batch 32, conv_dim 128: 4465 GFLOP/img/step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pair = (sys.argv[3] != 'nopair') if len(sys.argv) > 3 else True
torch.manual_seed(5)
s = FCDSolver(None, image_size=256, g_conv_dim=128, d_conv_dim=128, batch_size=B, precision=prec).to(dev)
s.pair_d_calls = s.pair_g_calls = pair
batch, _ = synth_batch(B, 256, 10, seed=3); b = tuple(t.to(dev) for t in batch)
for _ in range(2): s.train_step(b, force_g=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 4
for _ in range(n): s.train_step(b, force_g=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"stress {prec} B={B} pair={pair and s._pair_fits(b[0], s.g_conv_dim)}: {dt * 1e3:.1f} ms/step, {B / dt:.1f} images/s, {4465e9 * B / dt / 1e12:.1f} TFLOP/s, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
