"""D-step at full size (the parity test's states) in fp32x3 and fp32x2, twice each with allocator poisoning between: are the
discriminator's gradients reproducible, and the same in both modes (its kernels are BF16X3 in both)?"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import fcd_oracle as O
from fcd_amd import FCDSolver
dev = torch.device('cuda:0')
gs = O.synth_generator_state(); ds = O.synth_discriminator_state()
batch, alpha = O.synth_batch(16, 128, 10)
b = tuple(t.to(dev) for t in batch)


def poison(val):
    ts = [torch.full((n,), val, device=dev) for n in (1 << 28, 1 << 27, 1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18) for _ in range(3)]
    torch.cuda.synchronize()
    del ts


LGP = float(os.environ.get("LGP", "10"))


def dstep(mode):
    s = FCDSolver(None, precision=mode, lambda_gp=LGP)
    s.G.load_state_dict(gs); s.D.load_state_dict(ds); s.to(dev)
    s.alpha_override = alpha
    s.G.requires_grad_(False)
    s.training_step(b, 0, 0).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in s.D.named_parameters()}


runs = {}
for i, mode in enumerate(("fp32x3", "fp32x2", "fp32x3", "fp32x2")):
    poison(float("nan") if i % 2 else 3.0e30)
    runs[(mode, i // 2)] = dstep(mode)
for k in runs[("fp32x3", 0)]:
    a0, a1, c0, c1 = (runs[(m, j)][k].double() for m in ("fp32x3", "fp32x2") for j in (0, 1))
    n = float(a0.norm()) + 1e-300
    print(f"{k:22s} x3 run0 vs run1 {float((a0 - a1).norm()) / n:.2e}   x2 run0 vs run1 {float((c0 - c1).norm()) / n:.2e}   x2 vs x3 {float((a0 - c0).norm()) / n:.2e}")
