"""Where the host's share of a training step goes: cProfile over the enqueue loop of the full-size step (the GPU runs
behind it; no synchronisation inside the profiled region except what the step itself does).
python tools/host_profile.py [steps]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(8888)
s = FCDSolver(None, batch_size=16, precision='fp32x2').to(dev)
batch, _ = synth_batch(16, 128, 10, seed=8888); b = tuple(t.to(dev) for t in batch)
for _ in range(10): s.train_step(b, force_g=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): s.train_step(b, force_g=True)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"enqueue {t_enq / steps * 1e3:.2f} ms/step, with sync {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps): s.train_step(b, force_g=True)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
