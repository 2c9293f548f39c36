"""Module- and step-level parity of the HIP path against the golden fixtures written by
oracle/make_golden.py from the reference's own Generator/Discriminator (tiny config), and
against the CPU oracle on the same inputs.

fp32 bar: rel-err <= 1e-4 per tensor on losses and gradients (BASELINE.json north_star).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import relerr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


@pytest.fixture(scope="module")
def tiny():
    z = np.load(os.path.join(GOLD, "tiny_step.npz"))
    sc = json.load(open(os.path.join(GOLD, "tiny_scalars.json")))
    return z, sc


def build(z, cfg, dev):
    from fcd_amd import FCDSolver
    s = FCDSolver(None, c_dim=cfg["c_dim"], image_size=cfg["size"], g_conv_dim=cfg["g_conv_dim"],
                  d_conv_dim=cfg["d_conv_dim"], g_repeat_num=cfg["g_repeat"], d_repeat_num=cfg["d_repeat"],
                  num_channels=cfg["bands"], batch_size=cfg["batch"])
    gs = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("G.")}
    ds = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("D.")}
    s.G.load_state_dict(gs)
    s.D.load_state_dict(ds)
    s.to(dev)
    return s


def batch_of(z, dev):
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    return (t("x"), t("c_org"), t("c_trg"), t("label_org"), t("label_trg"))


def test_generator_forward_train_eval(dev, tiny):
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    b = batch_of(z, dev)
    s.G.train()
    with torch.no_grad():
        out = s.G(b[0], b[2])
    assert relerr(out, torch.from_numpy(z["G_train_out"])) <= TOL
    # invalid pixels pass through exactly
    inv = (b[0] == -1)
    assert inv.any() and bool((out[inv] == -1).all())
    for k in z.files:
        if k.startswith("G_train_stats."):
            name = k[len("G_train_stats."):]
            assert relerr(s.G.state_dict()[name], torch.from_numpy(z[k])) <= TOL, name
    s2 = build(z, sc["cfg"], dev)
    s2.G.eval()
    with torch.no_grad():
        out = s2.G(b[0], torch.zeros_like(b[2]))
    assert relerr(out, torch.from_numpy(z["G_eval_out"])) <= TOL


def test_discriminator_forward(dev, tiny):
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    b = batch_of(z, dev)
    with torch.no_grad():
        src, cls, feat = s.D(b[0])
    assert src.shape == z["D_src"].shape and cls.shape == z["D_cls"].shape
    assert relerr(src, torch.from_numpy(z["D_src"])) <= TOL
    assert relerr(cls, torch.from_numpy(z["D_cls"])) <= TOL
    assert relerr(feat[0], torch.from_numpy(z["D_feat"])) <= TOL


def test_d_step_losses_and_grads(dev, tiny):
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    b = batch_of(z, dev)
    s.alpha_override = torch.from_numpy(z["alpha"])
    s.G.requires_grad_(False)
    loss = s.training_step(b, 0, 0)
    loss.backward()
    torch.cuda.synchronize()
    for k, v in sc["d_losses"].items():
        assert abs(float(s.loss_dict[k]) - v) <= TOL * max(1.0, abs(v)), k
    assert abs(float(loss) - sc["d_loss"]) <= TOL * max(1.0, abs(sc["d_loss"]))
    worst = 0.0
    for name, p in s.D.named_parameters():
        ref = torch.from_numpy(z["dgrad." + name])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        e = relerr(got, ref)
        worst = max(worst, e)
        assert e <= TOL, (name, e)
    for k in z.files:
        if k.startswith("stats_after_d."):
            name = k[len("stats_after_d."):]
            assert relerr(s.G.state_dict()[name], torch.from_numpy(z[k])) <= TOL, name
    print("D-step worst grad rel-err", worst)


def test_g_step_losses_and_grads(dev, tiny):
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    b = batch_of(z, dev)
    s.D.requires_grad_(False)
    s.global_step = s.n_critic - 1
    loss = s.training_step(b, 0, 1)
    loss.backward()
    torch.cuda.synchronize()
    for k, v in sc["g_losses"].items():
        assert abs(float(s.loss_dict[k]) - v) <= TOL * max(1.0, abs(v)), k
    assert abs(float(loss) - sc["g_loss"]) <= TOL * max(1.0, abs(sc["g_loss"]))
    worst = 0.0
    for name, p in s.G.named_parameters():
        ref = torch.from_numpy(z["ggrad." + name])
        e = relerr(p.grad, ref)
        worst = max(worst, e)
        assert e <= TOL, (name, e)
    for k in z.files:
        if k.startswith("stats_after_g."):
            name = k[len("stats_after_g."):]
            assert relerr(s.G.state_dict()[name], torch.from_numpy(z[k])) <= TOL, name
    print("G-step worst grad rel-err", worst)


def test_masks_bit_exact_from_same_diff(dev, tiny):
    """Mask == (diff > thr) in fp32 on the diff map the same launch produced, and the diff
    map matches the reference's within tolerance."""
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    b = batch_of(z, dev)
    s.G.eval()
    diff, mask = s.difference_and_mask(b[0], sc["thr"])
    assert relerr(diff, torch.from_numpy(z["diff_eval"])) <= TOL
    ref_mask = (diff.cpu() > sc["thr"]).to(torch.uint8)        # torch's own fp32 compare
    assert bool((mask.cpu() == ref_mask).all())
    # away from the threshold the masks equal the reference's
    far = (torch.from_numpy(z["diff_eval"]) - sc["thr"]).abs() > 1e-5
    assert bool((mask.cpu()[far] == torch.from_numpy(z["mask_eval"])[far]).all())
