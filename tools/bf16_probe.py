"""Measure bf16-variant errors of the tiny-config D and G steps against the golden fixtures."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_step_gpu import build, batch_of, GOLD
from conftest import relerr
dev = torch.device("cuda", 0)
z = np.load(os.path.join(GOLD, "tiny_step.npz")); sc = json.load(open(os.path.join(GOLD, "tiny_scalars.json")))
for which in ("d", "g"):
    s = build(z, sc["cfg"], dev); s.set_precision("bf16"); b = batch_of(z, dev)
    if which == "d":
        s.alpha_override = torch.from_numpy(z["alpha"]); s.G.requires_grad_(False)
        loss = s.training_step(b, 0, 0); net, key, ld = s.D, "dgrad.", sc["d_losses"]
    else:
        s.D.requires_grad_(False); s.global_step = s.n_critic - 1
        loss = s.training_step(b, 0, 1); net, key, ld = s.G, "ggrad.", sc["g_losses"]
    loss.backward(); torch.cuda.synchronize()
    for k, v in ld.items():
        print(which, "loss", k, float(s.loss_dict[k]), v, abs(float(s.loss_dict[k]) - v) / max(1.0, abs(v)))
    worst = 0
    for name, p in net.named_parameters():
        e = relerr(p.grad if p.grad is not None else torch.zeros_like(p), torch.from_numpy(z[key + name])); worst = max(worst, e)
        print(which, "grad", name, "%.3e" % e)
    print(which, "worst", worst)
