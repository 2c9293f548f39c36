import json, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_step_gpu import build, batch_of, GOLD
from conftest import relerr
from fcd_amd import losses
dev = torch.device("cuda", 0)
z = np.load(os.path.join(GOLD, "tiny_step.npz")); sc = json.load(open(os.path.join(GOLD, "tiny_scalars.json")))
b = None
def run(arena, which):
    s = build(z, sc["cfg"], dev)
    if arena: s.optimizers()
    b = batch_of(z, dev)
    s.G.requires_grad_(False)
    x = b[0]
    if which == "real":
        out_src, out_cls, _ = s.D(x); loss = losses.mean_scaled(out_src, -1.0)
    elif which == "cls":
        out_src, out_cls, _ = s.D(x); loss = s.classification_loss(out_cls, b[3])
    elif which == "two":
        o1, _, _ = s.D(x); o2, _, _ = s.D(x * 0.5); loss = losses.mean_scaled(o1, -1.0) + losses.mean_scaled(o2, 1.0)
    else:
        xh = (x * 0.9).detach().requires_grad_(True)
        out_src, _, _ = s.D(xh); loss = s.gradient_penalty(out_src, xh)
    loss.backward(); torch.cuda.synchronize()
    return {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in s.D.named_parameters()}
for which in ("real", "cls", "two", "gp"):
    a, r = run(True, which), run(False, which)
    print(which, {n: "%.2e" % relerr(a[n], r[n]) for n in a})
