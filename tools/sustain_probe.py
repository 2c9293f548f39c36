"""Step time over a few hundred steps: does the short bench window hold under sustained load?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32x3'
if len(sys.argv) > 2:      # one big cached segment for the allocator to carve from
    _pre = torch.empty(int(sys.argv[2]) << 30, dtype=torch.uint8, device=dev); del _pre
torch.manual_seed(8888)
solver = FCDSolver(None, batch_size=16, precision=prec).to(dev)
batch, _ = synth_batch(16, 128, 10, seed=8888); batch = tuple(t.to(dev) for t in batch)
for _ in range(3): solver.train_step(batch, force_g=True)
torch.cuda.synchronize(); t00 = time.perf_counter()
out = []
for blk in range(8):
    t0 = time.perf_counter()
    for _ in range(20): solver.train_step(batch, force_g=True)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 20 * 1e3, 1))
print(prec, 'ms/step per 20-step block:', out, 'overall', round((time.perf_counter() - t00) / 160 * 1e3, 2), flush=True)
