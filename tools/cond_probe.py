"""How much does the full-size D-step gradient error (vs the fp64 oracle) move under mathematically
neutral changes of the fp32 summation order?  Permuting the batch changes the order in which the
weight-gradient kernels add up pixels, nothing else."""
import json, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import fcd_oracle as O
from fcd_amd import FCDSolver
GOLD = os.path.join(R, "tests", "golden")
dev = torch.device("cuda", 0)
sc = json.load(open(os.path.join(GOLD, "full_scalars.json"))); cfg = sc["cfg"]
BIG = np.load(os.path.join(GOLD, "full_grad_samples.npz"))
gs = O.synth_generator_state(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
ds = O.synth_discriminator_state(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"])
batch, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])
def errs(s):
    out = {}
    for name, p in s.D.named_parameters():
        idx = torch.from_numpy(BIG[f"d.idx.{name}"]); o64 = torch.from_numpy(BIG[f"d.ora64.{name}"]).double()
        r32 = torch.from_numpy(BIG[f"d.ref32.{name}"]).double()
        got = p.grad.detach().flatten().cpu()[idx].double(); den = float(o64.norm()) + 1e-30
        out[name] = (float((got - o64).norm()) / den, float((r32 - o64).norm()) / den)
    return out
g = torch.Generator().manual_seed(0)
for prec in ("fp32", "fp32x3"):
    for trial in range(5):
        perm = torch.arange(cfg["batch"]) if trial == 0 else torch.randperm(cfg["batch"], generator=g)
        s = FCDSolver(None, precision=prec); s.G.load_state_dict(gs); s.D.load_state_dict(ds); s.to(dev)
        b = tuple(t[perm].contiguous().to(dev) for t in batch)
        s.alpha_override = alpha[perm].contiguous()
        s.G.requires_grad_(False)
        s.training_step(b, 0, 0).backward()
        e = errs(s)
        worst = max(e.items(), key=lambda kv: kv[1][0] / max(1e-4, 2 * kv[1][1]))
        print(prec, "perm" if trial else "id  ", " ".join(f"{k.split('.')[0][:4]}{k.split('.')[1]}{'w' if 'weight' in k else 'b'}:{v[0] / max(v[1], 1e-30):.2f}" for k, v in e.items() if 'weight' in k),
              "| worst/bar %.2f (%s)" % (worst[1][0] / max(1e-4, 2 * worst[1][1]), worst[0]), flush=True)
