"""Generator forward (training mode, no grad: the D-step's fake images) in the three fp32-faithful modes against the
float64 oracle's forward: where does each mode's error sit?"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import fcd_oracle as O
from fcd_amd import FCDSolver
dev = torch.device('cuda:0')
gs = O.synth_generator_state(); ds = O.synth_discriminator_state()
batch, alpha = O.synth_batch(16, 128, 10)
b = tuple(t.to(dev) for t in batch)
out = {}
for mode in ("fp32", "fp32x3", "fp32x2"):
    s = FCDSolver(None, precision=mode)
    s.G.load_state_dict(gs); s.D.load_state_dict(ds); s.to(dev)
    s.G.train()
    with torch.no_grad():
        out[mode] = s.G(b[0], b[2]).double().cpu()
try:
    gs64 = O.to_dtype(gs, torch.float64)
    r = O.generator_forward(gs64, batch[0].double(), batch[2].double(), training=True)
    ref = (r[0] if isinstance(r, (tuple, list)) else r).detach()
except Exception as e:
    print("no float64 forward:", e); ref = None
for m in out:
    line = f"{m:7s}"
    for k in out:
        line += f"  vs {k} {float((out[m] - out[k]).norm() / out[k].norm()):.2e} (max {float((out[m] - out[k]).abs().max()):.2e})"
    if ref is not None:
        line += f"  vs float64 {float((out[m] - ref).norm() / ref.norm()):.2e}"
    print(line)
