"""BASELINE.json configs[4], the scaled stress configuration (image_size 256, batch 32 per GPU,
conv_dim widened to 128: 142.9 TFLOP per G+D step), through size-independent properties -- the
CPU oracle would need many minutes per step at this size.  Checks that every kernel's 32-bit
offset / tile-grid assumptions hold at 4x the pixels and 2x the channels of the baseline."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stress_config_properties(dev):
    from fcd_amd import FCDSolver
    from fcd_amd.synth import synth_batch
    torch.manual_seed(5)
    s = FCDSolver(None, image_size=256, g_conv_dim=128, d_conv_dim=128, batch_size=32, precision="fp32x3").to(dev)
    batch, _ = synth_batch(32, 256, 10, seed=3)
    b = tuple(t.to(dev) for t in batch)
    x = b[0]
    s.G.train()
    with torch.no_grad():
        o32 = s.G(x, b[2])
        o4 = s.G(x[4:8].contiguous(), b[2][4:8].contiguous())
        src, cls, feat = s.D(x)
    assert o32.shape == x.shape and torch.isfinite(o32).all()
    # per-sample independence (InstanceNorm, what data parallelism relies on)
    assert float((o32[4:8] - o4).abs().max()) <= 2e-5
    inv = x == -1
    assert bool(inv.any()) and bool((o32[inv] == -1).all())
    assert src.shape == (32, 1, 4, 4) and cls.shape == (32, 1) and feat[0].shape == (32, 1024, 16, 16)
    # D is piecewise linear in x: its input gradient is linear in grad_outputs
    xr = x[:8].clone().requires_grad_(True)
    sr, _, _ = s.D(xr)
    g1 = torch.autograd.grad(sr, xr, torch.ones_like(sr), retain_graph=True)[0]
    g3 = torch.autograd.grad(sr, xr, 3 * torch.ones_like(sr))[0]
    assert float((g3 - 3 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())
    # exact-fp32 and fp32x3 agree on the forward pass at this size
    s.set_precision("fp32")
    with torch.no_grad():
        o32e = s.G(x, b[2])
    s.set_precision("fp32x3")
    assert float((o32e - o32).double().norm() / o32e.double().norm()) <= 1e-5
    # two full train steps: finite losses, every weight moved by about lr
    w0 = s.G.main[0].weight.detach().clone()
    d0 = s.D.conv1.weight.detach().clone()
    for _ in range(2):
        d_loss, g_loss = s.train_step(b, force_g=True)
    assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
    assert 1e-5 < float((s.G.main[0].weight - w0).abs().max()) < 1e-3
    assert 1e-5 < float((s.D.conv1.weight - d0).abs().max()) < 1e-3
    # masks: bit-exact against torch's fp32 comparison of the same difference map
    diff, mask = s.difference_and_mask(x, 0.027)
    assert torch.equal(mask.cpu(), (diff.cpu() > 0.027).to(torch.uint8))
