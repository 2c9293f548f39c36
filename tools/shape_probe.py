"""Per-tensor G-step gradient errors of the HIP path and of the fp32 oracle against the fp64 oracle
for one of tests/test_step_gpu.py's SHAPES (is a parity miss a bug or conditioning?)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_amd import FCDSolver
from oracle import fcd_oracle as O
b, size, bands, c_dim, gcd, dcd, grep, drep = [int(v) for v in sys.argv[1:9]]
dev = torch.device('cuda:0')
gs = O.synth_generator_state(gcd, c_dim, grep, bands)
ds = O.synth_discriminator_state(size, dcd, c_dim, drep, bands)
(x, _, _, _, _), alpha = O.synth_batch(b, size, bands, seed=11)
rng = np.random.default_rng(12)
lo = torch.from_numpy((rng.random((b, c_dim)) < 0.5).astype(np.float32))
lt = lo[torch.from_numpy(rng.permutation(b))]
batch = (x, lo.clone(), lt.clone(), lo, lt)
_, gl, gg = O.g_step(gs, ds, batch, new_stats={})
b64 = tuple(t.double() for t in batch)
_, gl64, gg64 = O.g_step(O.to_dtype(gs, torch.float64), O.to_dtype(ds, torch.float64), b64, new_stats={})
s = FCDSolver(None, c_dim=c_dim, image_size=size, g_conv_dim=gcd, d_conv_dim=dcd, g_repeat_num=grep, d_repeat_num=drep,
              num_channels=bands, batch_size=b)
s.G.load_state_dict(gs); s.D.load_state_dict(ds); s.to(dev)
s.D.requires_grad_(False); s.global_step = s.n_critic - 1
s.training_step(tuple(t.to(dev) for t in batch), 0, 1).backward()
rel = lambda a, r: float((a.double().cpu() - r.double()).norm() / (r.double().norm() + 1e-30))
for k, v in gl64.items():
    print(f"{k:18s} hip {float(s.loss_dict[k]):.7f} ora32 {float(gl[k]):.7f} ora64 {float(v):.7f}")
for name, p in s.G.named_parameters():
    print(f"{name:28s} hip-vs-64 {rel(p.grad, gg64[name]):.2e}  ora32-vs-64 {rel(gg[name], gg64[name]):.2e}  hip-vs-ora32 {rel(p.grad, gg[name]):.2e}")
