import json, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_step_gpu import build, batch_of, GOLD, _oracle_two_steps
from conftest import relerr
from oracle import fcd_oracle as O
dev = torch.device("cuda", 0)
z = np.load(os.path.join(GOLD, "tiny_step.npz")); sc = json.load(open(os.path.join(GOLD, "tiny_scalars.json")))
lr = 1e-3
s = build(z, sc["cfg"], dev); s.d_lr = s.g_lr = lr
s.alpha_override = torch.from_numpy(z["alpha"]); b = batch_of(z, dev)
d0 = {k: v.clone() for k, v in s.D.state_dict().items()}
s.train_step(b, force_g=True)
print("losses step1", {k: float(v) for k, v in s.loss_dict.items()})
# oracle one step
gs = {k[2:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("G.")}
ds = {k[2:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("D.")}
t = lambda k: torch.from_numpy(z[k])
batch = (t("x"), t("c_org"), t("c_trg"), t("label_org"), t("label_trg"))
ns = {}
_, ld, dg = O.d_step(gs, ds, batch, t("alpha"), new_stats=ns)
print("oracle losses", {k: float(v) for k, v in ld.items()})
for k, g in dg.items():
    new, m, v = O.adam_update(ds[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr=lr)
    got = s.D.state_dict()[k].cpu()
    print(k, "param relerr %.3e" % relerr(got, new), "delta-norm hip %.3e oracle %.3e" % (float((got - d0[k].cpu()).norm()), float((new - ds[k]).norm())),
          "max|delta| hip %.3e" % float((got - d0[k].cpu()).abs().max()))
