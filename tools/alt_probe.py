import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
torch.manual_seed(8888)
solver = FCDSolver(None, batch_size=16, precision='fp32x3').to(dev)
batch, _ = synth_batch(16, 128, 10, seed=8888); batch = tuple(t.to(dev) for t in batch)
def run(prec, n=10):
    solver.set_precision(prec)
    for _ in range(3): solver.train_step(batch, force_g=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): solver.train_step(batch, force_g=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(prec, round(dt / n * 1e3, 2), 'ms/step', flush=True)
for p in ('fp32x3', 'fp32x3', 'fp32x3', 'fp32', 'fp32x3', 'fp32x3'):
    run(p)
