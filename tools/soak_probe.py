"""Soak + determinism: N training steps twice from the same state; every loss finite, and the two
runs' final weights and running statistics bit-identical (fixed-order reductions, no atomics in the
training path: any difference would point at a race in a kernel or in the stream ordering)."""
import os, sys, time, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from fcd_amd.synth import synth_batch
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32x3'
batches = [tuple(t.to(dev) for t in synth_batch(16, 128, 10, seed=100 + i)[0]) for i in range(4)]


def run():
    torch.manual_seed(8888)
    s = FCDSolver(None, batch_size=16, precision=prec).to(dev)
    s.alpha_override = torch.full((16, 1, 1, 1), 0.37)
    t0 = time.perf_counter()
    for i in range(steps):
        d_loss, g_loss = s.train_step(batches[i % 4], force_g=True)
        if i % 50 == 49:
            assert torch.isfinite(d_loss) and torch.isfinite(g_loss), (i, float(d_loss), float(g_loss))
            print(f"  step {i + 1}: d_loss {float(d_loss):+.5f} g_loss {float(g_loss):+.5f} "
                  f"({(time.perf_counter() - t0) / (i + 1) * 1e3:.1f} ms/step)", flush=True)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for net in (s.G, s.D):
        for k, v in net.state_dict().items():
            h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


a = run()
b = run()
print("run 1", a[:16], "run 2", b[:16], "IDENTICAL" if a == b else "DIFFERENT", flush=True)
sys.exit(0 if a == b else 1)
