"""CPU suite: the oracle against the golden fixtures made from the reference's own modules
(oracle/make_golden.py), threshold semantics, scene tiling."""
import json
import os

import numpy as np
import torch

from conftest import relerr
from oracle import fcd_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_tiny():
    z = np.load(os.path.join(GOLD, "tiny_step.npz"))
    sc = json.load(open(os.path.join(GOLD, "tiny_scalars.json")))
    gs = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("G.")}
    ds = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("D.")}
    t = lambda k: torch.from_numpy(z[k])
    batch = (t("x"), t("c_org"), t("c_trg"), t("label_org"), t("label_trg"))
    return z, sc, gs, ds, batch


def test_synth_recipe_is_reproducible():
    z, sc, gs, ds, batch = load_tiny()
    cfg = sc["cfg"]
    b2, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])
    assert torch.equal(b2[0], batch[0]) and torch.equal(alpha, torch.from_numpy(z["alpha"]))
    g2 = O.synth_generator_state(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
    assert all(torch.equal(g2[k], gs[k]) for k in gs)
    # the nodata rectangle maps to exactly -1
    assert bool((batch[0][0, :, :4, :8] == -1).all())


def test_oracle_forward_matches_reference_fixtures():
    z, sc, gs, ds, batch = load_tiny()
    ns = {}
    with torch.no_grad():
        out = O.generator_forward(gs, batch[0], batch[2], True, "relu", ns)
    assert relerr(out, torch.from_numpy(z["G_train_out"])) <= 2e-6
    for k in z.files:
        if k.startswith("G_train_stats."):
            assert relerr(ns[k[len("G_train_stats."):]], torch.from_numpy(z[k])) <= 2e-6
    with torch.no_grad():
        ev = O.generator_forward(gs, batch[0], torch.zeros(batch[0].shape[0], 1), False)
        src, cls, feat = O.discriminator_forward(ds, batch[0])
    assert relerr(ev, torch.from_numpy(z["G_eval_out"])) <= 2e-6
    assert relerr(src, torch.from_numpy(z["D_src"])) <= 2e-6
    assert relerr(cls, torch.from_numpy(z["D_cls"])) <= 2e-6
    assert relerr(feat[0], torch.from_numpy(z["D_feat"])) <= 2e-6


def test_oracle_steps_match_reference_fixtures():
    z, sc, gs, ds, batch = load_tiny()
    alpha = torch.from_numpy(z["alpha"])
    loss, losses, grads = O.d_step(gs, ds, batch, alpha, new_stats={})
    for k, v in sc["d_losses"].items():
        assert abs(float(losses[k]) - v) <= 1e-5 * max(1, abs(v))
    for k, g in grads.items():
        assert relerr(g, torch.from_numpy(z["dgrad." + k])) <= 1e-4, k
    ns = {}
    loss, losses, grads = O.g_step(gs, ds, batch, new_stats=ns)
    for k, v in sc["g_losses"].items():
        assert abs(float(losses[k]) - v) <= 1e-5 * max(1, abs(v))
    for k, g in grads.items():
        assert relerr(g, torch.from_numpy(z["ggrad." + k])) <= 1e-4, k
    for k in z.files:
        if k.startswith("stats_after_g."):
            assert relerr(ns[k[len("stats_after_g."):]], torch.from_numpy(z[k])) <= 2e-6


def test_oracle_masks_and_scalars():
    z, sc, gs, ds, batch = load_tiny()
    d = O.difference_map(gs, batch[0], False)
    assert relerr(d, torch.from_numpy(z["diff_eval"])) <= 2e-6
    m = O.threshold_mask(torch.from_numpy(z["diff_eval"]), sc["thr"])
    assert torch.equal(m, torch.from_numpy(z["mask_eval"]))
    assert int(m.sum()) == sc["mask_popcount"]
    pm = O.pseudo_mask(torch.from_numpy(z["diff_eval"]), sc["thr"], batch[0][:, 0] != -1)
    assert set(np.unique(pm.numpy()).tolist()) <= {0, 1, 2} and bool((pm[batch[0][:, 0] == -1] == 0).all())


def test_threshold_semantics():
    cases = json.load(open(os.path.join(GOLD, "threshold_cases.json")))
    for thr_s, c in cases.items():
        thr = float(thr_s)
        vals = np.array(c["vals_bits"], dtype=np.uint32).view(np.float32)
        assert O.threshold_mask(torch.from_numpy(vals.copy()), thr).tolist() == c["torch_gt"]
        assert O.binarize(vals.astype(np.float64), thr).tolist() == c["numpy_f64_gt"]
    # fp32(0.027) > 0.027 in double, yet the fp32 compare says "not greater"
    c = cases["0.027"]
    assert c["torch_gt"][1] == 0 and c["numpy_f64_gt"][1] == 1


def test_scene_tiling_matches_reference_fixture():
    z = np.load(os.path.join(GOLD, "tiny_scene.npz"))
    gs = O.synth_generator_state(8, 1, 2, 10)
    diff = O.predict_scene(gs, z["scene"], 32, 2, False)
    assert diff.dtype == np.float64 and diff.shape == z["diff"].shape
    assert np.abs(diff - z["diff"]).max() <= 1e-6
    far = np.abs(z["diff"] - 0.02) > 1e-6
    assert np.array_equal(O.binarize(diff, 0.02)[far], z["mask_0.02"][far])
    origins, pad_h, pad_w, cps = O.scene_patches(75, 52, 32)
    assert origins[:3] == [(0, 0), (0, 32), (32, 0)] and pad_h == (0, 32 - 75 % 32) and cps == 32


def test_adam_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(100)
    g1, g2 = torch.randn(100), torch.randn(100)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], 1e-4, (0.5, 0.999))
    m, v, q = torch.zeros(100), torch.zeros(100), p.clone()
    for step, g in enumerate((g1, g2), 1):
        ref.grad = g.clone()
        opt.step()
        q, m, v = O.adam_update(q, g, m, v, step)
    assert relerr(q, ref.detach()) <= 1e-6


def test_full_size_fixture_is_pinned():
    sc = json.load(open(os.path.join(GOLD, "full_scalars.json")))
    assert sc["cfg"]["batch"] == 16 and sc["cfg"]["size"] == 128
    assert sc["oracle_fp64_vs_reference_fp64_worst_relerr"] <= 1e-9


def test_oracle_variants_match_reference_fixtures():
    """use_feats / interm_non_act / ReLU discriminator: the oracle against
    the reference-generated tests/golden/tiny_variants.npz (make_golden.py --variants)."""
    import re
    import pytest
    variants = {"feats": ("relu", "lrelu", 2, False, True), "non_act": ("relu", "lrelu", 2, True, True),
                "non_act_allfeat": ("relu", "lrelu", 4, True, True), "relu_d": ("lrelu", "relu", 2, False, False),
                "silu_g": ("silu", "lrelu", 4, False, False), "mish_g": ("mish", "lrelu", 4, False, True),
                "silu_d": ("relu", "silu", 2, False, True), "mish_d": ("silu", "mish", 2, True, True)}
    z, sc, gs, _, _ = load_tiny()
    cfg = sc["cfg"]
    v = np.load(os.path.join(GOLD, "tiny_variants.npz"))
    batch, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])
    for name, (act_g, act_d, n_feat, ina, feats) in variants.items():
        ds = O.synth_discriminator_state(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"],
                                         n_feature_layers=n_feat)
        rk = (lambda k: re.sub(r"^(features|main)\.(\d+)\.0\.", r"\1.\2.1.", k)) if ina else (lambda k: k)
        with torch.no_grad():
            src, _, feat = O.discriminator_forward(ds, batch[0], act_d, ina)
        assert relerr(src, torch.from_numpy(v[f"{name}.D_src"])) <= 2e-6
        assert relerr(feat[0], torch.from_numpy(v[f"{name}.D_feat"])) <= 2e-6
        _, dl, dg = O.d_step(gs, ds, batch, alpha, act_g=act_g, act_d=act_d, new_stats={}, interm_non_act=ina)
        for k, t in dl.items():
            assert float(t) == pytest.approx(float(v[f"{name}.dloss.{k}"]), rel=1e-5, abs=1e-5)
        for k, g in dg.items():
            assert relerr(g, torch.from_numpy(v[f"{name}.dgrad.{rk(k)}"])) <= 1e-4, (name, k)
        _, gl, gg = O.g_step(gs, ds, batch, act_g=act_g, act_d=act_d, new_stats={}, use_feats=feats, interm_non_act=ina)
        assert len(gl) == (10 if feats else 7)
        for k, t in gl.items():
            assert float(t) == pytest.approx(float(v[f"{name}.gloss.{k}"]), rel=1e-5, abs=1e-5)
        for k, g in gg.items():
            assert relerr(g, torch.from_numpy(v[f"{name}.ggrad.{k}"])) <= 1e-4, (name, k)
