"""Host cost of a training step: the same launch sequence on a tiny configuration (GPU work negligible)
is host-bound, so its step time is what the host needs to enqueue a step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
torch.manual_seed(8888)
s = FCDSolver(None, image_size=64, g_conv_dim=8, d_conv_dim=8, batch_size=2, precision='fp32x3').to(dev)
batch, _ = synth_batch(2, 64, 10, seed=8888); b = tuple(t.to(dev) for t in batch)
for _ in range(10): s.train_step(b, force_g=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): s.train_step(b, force_g=True)
torch.cuda.synchronize()
print(f"tiny configuration (same layer count, 6 res blocks, D x6... image 64, conv_dim 8, B=2): {(time.perf_counter() - t0) / 50 * 1e3:.1f} ms/step", flush=True)
