"""Does the fp32x3 G-step's result depend on what ran before it in the process?  (hash of the generator gradients)"""
import hashlib, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import fcd_oracle as O
from fcd_amd import FCDSolver
dev = torch.device("cuda:0")
gs = O.synth_generator_state(); ds = O.synth_discriminator_state()
batch, alpha = O.synth_batch(16, 128, 10)
b = tuple(t.to(dev) for t in batch)


def gstep(prec):
    s = FCDSolver(None, precision=prec)
    s.G.load_state_dict(gs); s.D.load_state_dict(ds); s.to(dev)
    s.D.requires_grad_(False); s.global_step = s.n_critic - 1
    s.training_step(b, 0, 1).backward()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for p in s.G.parameters():
        h.update(p.grad.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def poison(val):
    """fill ~40 GB of allocator blocks of assorted sizes with `val`, then hand them back to the caching allocator"""
    ts = [torch.full((n,), val, device=dev) for n in (1 << 28, 1 << 28, 1 << 27, 1 << 26, 1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18)
          for _ in range(3)]
    torch.cuda.synchronize()
    del ts


seq = []
for a in sys.argv[1:]:
    if a.startswith("poison"):
        poison(float(a.split("=")[1]))
    else:
        seq.append(a)
print(os.environ.get("FCD_NO_HALO12"), os.environ.get("FCD_WGRAD_OLD"), sys.argv[1:], [gstep(p) for p in seq], flush=True)
