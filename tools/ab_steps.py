"""A/B step time of two builds of the library in one GPU session: FCD_HIP_LIB selects the build per
child process; builds alternate so that clock drift shows up as spread, not as a difference."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
code = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
torch.manual_seed(8888)
s = FCDSolver(None, batch_size=16, precision='fp32x3').to(dev)
batch, _ = synth_batch(16, 128, 10, seed=8888); b = tuple(t.to(dev) for t in batch)
for _ in range(25): s.train_step(b, force_g=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): s.train_step(b, force_g=True)
torch.cuda.synchronize()
print(round((time.perf_counter() - t0) / 40 * 1e3, 3))
''' % root
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ, FCD_HIP_LIB=os.path.join(root, "fcd_amd", lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        print(rnd, lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
