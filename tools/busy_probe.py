"""GPU busy fraction of the training step: run under `rocprofv3 --kernel-trace --stats` and compare the
summed kernel time of N steps with their wall time (printed here)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32x3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(8888)
solver = FCDSolver(None, batch_size=16, precision=prec).to(dev)
batch, _ = synth_batch(16, 128, 10, seed=8888); batch = tuple(t.to(dev) for t in batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): solver.train_step(batch, force_g=True)
torch.cuda.synchronize()
print(prec, steps, 'steps incl. the first, wall ms', round((time.perf_counter() - t0) * 1e3, 1), flush=True)
