"""Pin the oracle against the reference and write tests/golden/*.

Runs ONLY in the build container (needs /root/reference, read-only).  It
imports the reference's own ``models/fixed_point_gan.py`` (numpy + torch only),
drives ``Generator`` / ``Discriminator`` with the loss arithmetic of
``fcd_solver.py:227-239, 260-263, 286-302, 316-359, 404-406, 802-807`` (the
solver module itself needs Lightning/cv2/... and cannot be imported), compares
with ``oracle/fcd_oracle.py`` and stores the REFERENCE-side results as fixtures.
Fixtures are data (inputs / expected outputs); no reference source is stored.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--full]
"""
import argparse
import json
import os
import re
import sys
import zlib

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from models.fixed_point_gan import Generator, Discriminator  # noqa: E402  (the reference)
from oracle import fcd_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
LAM = O.DEFAULT_LAMBDAS


def ref_modules(gs, ds, cfg, act_g="relu", act_d="lrelu", n_feat=4, interm_non_act=False):
    G = Generator(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"], activation=act_g)
    D = Discriminator(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"], activation=act_d,
                      n_feature_layers=n_feat, interm_non_act=interm_non_act)
    G.load_state_dict(gs)
    D.load_state_dict(ds)
    return G, D


def ref_gp(y, x):
    dydx = torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True, create_graph=True)[0]
    dydx = dydx.view(dydx.size(0), -1)
    return torch.mean((torch.sqrt(torch.sum(dydx ** 2, dim=1)) - 1) ** 2)


def ref_d_step(G, D, batch, alpha):
    x_real, c_org, c_trg, label_org, label_trg = batch
    G.requires_grad_(False); D.requires_grad_(True)
    G.train(); D.train()
    D.zero_grad(set_to_none=True)
    out_src, out_cls, _ = D(x_real)
    l_real = -torch.mean(out_src)
    l_cls = F.binary_cross_entropy_with_logits(out_cls, label_org)
    x_fake = G(x_real, c_trg)
    out_src, out_cls, _ = D(x_fake.detach())
    l_fake = torch.mean(out_src)
    x_hat = (alpha * x_real.data + (1 - alpha) * x_fake.data).requires_grad_(True)
    out_src, _, _ = D(x_hat)
    l_gp = ref_gp(out_src, x_hat)
    loss = l_real + l_fake + LAM["cls"] * l_cls + LAM["gp"] * l_gp
    loss.backward()
    grads = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in D.named_parameters()}
    losses = {"D/loss_real": l_real, "D/loss_fake": l_fake, "D/loss_cls": l_cls, "D/loss_gp": l_gp}
    return loss.detach(), {k: v.detach() for k, v in losses.items()}, grads


def ref_g_step(G, D, batch, use_feats=False, lambda_feat=1.0):
    x_real, c_org, c_trg, label_org, label_trg = batch
    G.requires_grad_(True); D.requires_grad_(False)
    G.train(); D.train()
    G.zero_grad(set_to_none=True)
    x_fake = G(x_real, c_trg)
    out_src, out_cls, _ = D(x_fake)
    l_fake = -torch.mean(out_src)
    l_cls = F.binary_cross_entropy_with_logits(out_cls, label_trg)
    x_fake_id = G(x_real, c_org)
    out_src_id, out_cls_id, x_fake_id_feats = D(x_fake_id)
    l_fake_id = -torch.mean(out_src_id)
    l_cls_id = F.binary_cross_entropy_with_logits(out_cls_id, label_org)
    l_id = torch.mean(torch.abs(x_real - x_fake_id))
    x_reconst = G(x_fake, c_org)
    l_rec = torch.mean(torch.abs(x_real - x_reconst))
    x_reconst_id = G(x_fake_id, c_org)
    _, _, x_reconst_id_feats = D(x_reconst_id)
    l_rec_id = torch.mean(torch.abs(x_real - x_reconst_id))
    losses = {"G/loss_fake": l_fake, "G/loss_rec": l_rec, "G/loss_cls": l_cls,
              "G/loss_fake_id": l_fake_id, "G/loss_rec_id": l_rec_id,
              "G/loss_cls_id": l_cls_id, "G/loss_id": l_id}
    l_feat = 0
    if use_feats:      # the arithmetic of fcd_solver.py:361-385 (use_mask False)
        def fl1(fa, fb):
            return torch.mean(torch.stack([torch.mean(torch.abs(a - b)) for a, b in zip(fa, fb)]))
        _, _, x_real_feats = D(x_real)
        l_id_feat = fl1(x_real_feats, x_fake_id_feats)
        _, _, x_reconst_feats = D(x_reconst)
        l_rec_feat = fl1(x_real_feats, x_reconst_feats)
        l_rec_id_feat = fl1(x_real_feats, x_reconst_id_feats)
        l_feat = lambda_feat * LAM["rec"] * l_rec_feat + lambda_feat * LAM["rec"] * l_rec_id_feat \
            + lambda_feat * LAM["id"] * l_id_feat
        losses.update({"G/loss_id_feat": l_id_feat, "G/loss_rec_feat": l_rec_feat, "G/loss_rec_id_feat": l_rec_id_feat})
    same = l_fake_id + LAM["rec"] * l_rec_id + LAM["cls"] * l_cls_id + LAM["id"] * l_id
    loss = l_fake + LAM["rec"] * l_rec + LAM["cls"] * l_cls + same + l_feat
    loss.backward()
    grads = {k: p.grad.clone() for k, p in G.named_parameters()}
    return loss.detach(), {k: v.detach() for k, v in losses.items()}, grads


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, a, b, tol):
    e = relerr(a, b)
    assert e <= tol, f"oracle != reference on {name}: rel-err {e:.3e} > {tol}"
    return e


def sample_idx(numel, k=64, seed=7):
    rng = np.random.default_rng(seed + numel % 1000003)
    return np.sort(rng.choice(numel, size=min(k, numel), replace=False))


def run_config(cfg, store_tensors, nsamp=64):
    gs = O.synth_generator_state(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
    ds = O.synth_discriminator_state(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"])
    batch, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])
    x = batch[0]
    out = {}
    worst = 0.0

    # ---- module forward parity (train + eval, running-stat update) -------------
    G, D = ref_modules(gs, ds, cfg)
    G.train()
    with torch.no_grad():
        ref_train = G(x, batch[2])
    ref_stats = {k: v.clone() for k, v in G.state_dict().items() if "running_" in k}
    ns = {}
    with torch.no_grad():
        ora_train = O.generator_forward(gs, x, batch[2], True, "relu", ns)
    worst = max(worst, check("G train", ora_train, ref_train, 2e-6))
    for k, v in ref_stats.items():
        worst = max(worst, check(k, ns[k], v, 2e-6))
    G2, _ = ref_modules(gs, ds, cfg)
    G2.eval()
    with torch.no_grad():
        ref_eval = G2(x, torch.zeros(x.shape[0], 1))
        ora_eval = O.generator_forward(gs, x, torch.zeros(x.shape[0], 1), False)
    worst = max(worst, check("G eval", ora_eval, ref_eval, 2e-6))
    with torch.no_grad():
        r_src, r_cls, r_feat = D(x)
        o_src, o_cls, o_feat = O.discriminator_forward(ds, x)
    worst = max(worst, check("D src", o_src, r_src, 2e-6))
    worst = max(worst, check("D cls", o_cls, r_cls, 2e-6))
    worst = max(worst, check("D feat", o_feat[0], r_feat[0], 2e-6))

    # ---- difference map + masks (eval mode, like make_psuedo_masks) ------------
    with torch.no_grad():
        ref_diff = torch.mean(torch.abs(ref_eval - x) / 2, dim=1)
    ora_diff = O.difference_map(gs, x, False)
    worst = max(worst, check("diff map", ora_diff, ref_diff, 2e-6))

    # ---- D step ----------------------------------------------------------------
    G, D = ref_modules(gs, ds, cfg)
    rl, rlosses, rgrads = ref_d_step(G, D, batch, alpha)
    stats_after_d = {k: v.clone() for k, v in G.state_dict().items() if "running_" in k}
    ns = {}
    ol, olosses, ograds = O.d_step(gs, ds, batch, alpha, new_stats=ns)
    for k in rlosses:
        worst = max(worst, check(k, olosses[k], rlosses[k], 1e-5))
    for k in rgrads:
        worst = max(worst, check("dD/" + k, ograds[k], rgrads[k], 2e-2))
    d_loss, d_losses, d_grads = rl, rlosses, rgrads

    # ---- G step (fresh modules: same weights, running stats from synth state) ---
    G, D = ref_modules(gs, ds, cfg)
    rl, rlosses, rgrads = ref_g_step(G, D, batch)
    stats_after_g = {k: v.clone() for k, v in G.state_dict().items() if "running_" in k}
    ns = {}
    ol, olosses, ograds = O.g_step(gs, ds, batch, new_stats=ns)
    for k in rlosses:
        worst = max(worst, check(k, olosses[k], rlosses[k], 1e-5))
    for k in rgrads:
        worst = max(worst, check("dG/" + k, ograds[k], rgrads[k], 2e-2))
    for k, v in stats_after_g.items():
        worst = max(worst, check("G-step " + k, ns[k], v, 2e-6))
    g_loss, g_losses, g_grads = rl, rlosses, rgrads

    # ---- fp64 oracle of the same steps: the "true" value both fp32 paths approximate ---
    gs64, ds64 = O.to_dtype(gs, torch.float64), O.to_dtype(ds, torch.float64)
    b64 = tuple(t.double() for t in batch)
    _, dl64, dg64 = O.d_step(gs64, ds64, b64, alpha.double(), new_stats={})
    _, gl64, gg64 = O.g_step(gs64, ds64, b64, new_stats={})
    # strongest pin: the reference's own modules in fp64 against the fp64 oracle
    G, D = ref_modules(gs, ds, cfg)
    G.double(); D.double()
    _, _, rdg64 = ref_d_step(G, D, b64, alpha.double())
    G, D = ref_modules(gs, ds, cfg)
    G.double(); D.double()
    _, _, rgg64 = ref_g_step(G, D, b64)
    pin64 = 0.0
    for k in rdg64:
        pin64 = max(pin64, check("fp64 dD/" + k, dg64[k], rdg64[k], 1e-9))
    for k in rgg64:
        pin64 = max(pin64, check("fp64 dG/" + k, gg64[k], rgg64[k], 1e-9))
    ref32_vs_64 = {"d": {k: relerr(d_grads[k], dg64[k]) for k in d_grads},
                   "g": {k: relerr(g_grads[k], gg64[k]) for k in g_grads}}

    thr = 0.02794
    mask = (ref_diff > thr).to(torch.uint8)

    if store_tensors:
        t = {}
        t["x"] = x.numpy(); t["alpha"] = alpha.numpy()
        for i, n in enumerate(["c_org", "c_trg", "label_org", "label_trg"]):
            t[n] = batch[i + 1].numpy()
        for k, v in gs.items():
            t["G." + k] = v.numpy()
        for k, v in ds.items():
            t["D." + k] = v.numpy()
        t["G_train_out"] = ref_train.numpy(); t["G_eval_out"] = ref_eval.numpy()
        for k, v in ref_stats.items():
            t["G_train_stats." + k] = v.numpy()
        t["D_src"] = r_src.numpy(); t["D_cls"] = r_cls.numpy(); t["D_feat"] = r_feat[0].numpy()
        t["diff_eval"] = ref_diff.numpy(); t["mask_eval"] = mask.numpy()
        t["d_loss"] = d_loss.numpy(); t["g_loss"] = g_loss.numpy()
        for k, v in d_losses.items():
            t["dloss." + k] = v.numpy()
        for k, v in g_losses.items():
            t["gloss." + k] = v.numpy()
        for k, v in d_grads.items():
            t["dgrad." + k] = v.numpy()
        for k, v in g_grads.items():
            t["ggrad." + k] = v.numpy()
        for k, v in stats_after_d.items():
            t["stats_after_d." + k] = v.numpy()
        for k, v in stats_after_g.items():
            t["stats_after_g." + k] = v.numpy()
        out["tensors"] = t
    scal = {"cfg": cfg, "thr": thr,
            "d_loss": float(d_loss), "g_loss": float(g_loss),
            "d_losses": {k: float(v) for k, v in d_losses.items()},
            "g_losses": {k: float(v) for k, v in g_losses.items()},
            "d_grad_norm": {k: float(v.double().norm()) for k, v in d_grads.items()},
            "g_grad_norm": {k: float(v.double().norm()) for k, v in g_grads.items()},
            "d_grad_samples": {k: {"idx": sample_idx(v.numel()).tolist(),
                                   "val": v.flatten()[sample_idx(v.numel())].double().tolist()}
                               for k, v in d_grads.items()},
            "g_grad_samples": {k: {"idx": sample_idx(v.numel()).tolist(),
                                   "val": v.flatten()[sample_idx(v.numel())].double().tolist()}
                               for k, v in g_grads.items()},
            "nsamp_big": nsamp,
            "d_losses64": {k: float(v) for k, v in dl64.items()},
            "g_losses64": {k: float(v) for k, v in gl64.items()},
            "d_grad_samples64": {k: v.flatten()[sample_idx(v.numel())].tolist() for k, v in dg64.items()},
            "g_grad_samples64": {k: v.flatten()[sample_idx(v.numel())].tolist() for k, v in gg64.items()},
            "d_grad_norm64": {k: float(v.norm()) for k, v in dg64.items()},
            "g_grad_norm64": {k: float(v.norm()) for k, v in gg64.items()},
            "ref_fp32_vs_fp64_grad_relerr": ref32_vs_64,
            "oracle_fp64_vs_reference_fp64_worst_relerr": pin64,
            "G_eval_out_norm": float(ref_eval.double().norm()),
            "G_train_out_norm": float(ref_train.double().norm()),
            "diff_sum": float(ref_diff.double().sum()),
            "mask_popcount": int(mask.sum()),
            "mask_crc32": int(zlib.crc32(mask.numpy().tobytes())),
            "diff_margin_count": int(((ref_diff - thr).abs() < 1e-6).sum()),
            "oracle_vs_reference_worst_relerr": worst}
    out["scalars"] = scal
    if nsamp > 64:
        # larger per-tensor samples (fp32 reference and fp64 oracle) for stable rel-err estimates
        big = {}
        for tag, g32, g64 in (("d", d_grads, dg64), ("g", g_grads, gg64)):
            for k in g32:
                idx = sample_idx(g32[k].numel(), nsamp)
                big[f"{tag}.idx.{k}"] = idx.astype(np.int64)
                big[f"{tag}.ref32.{k}"] = g32[k].flatten()[idx].numpy()
                big[f"{tag}.ora64.{k}"] = g64[k].flatten()[idx].numpy()
                # a second, independent sample of the same size (another seed)
                idx2 = sample_idx(g32[k].numel(), nsamp, seed=1234)
                big[f"{tag}.idx2.{k}"] = idx2.astype(np.int64)
                big[f"{tag}.ref32_2.{k}"] = g32[k].flatten()[idx2].numpy()
                big[f"{tag}.ora64_2.{k}"] = g64[k].flatten()[idx2].numpy()
        out["big_samples"] = big
        # element-wise companions of the norm-wise figures, over the whole tensors: ||g64||_inf and the
        # reference-fp32 maximum absolute error relative to it
        scal["grad_absmax64"] = {tag: {k: float(v.abs().max()) for k, v in g64.items()}
                                 for tag, g64 in (("d", dg64), ("g", gg64))}
        scal["ref_fp32_vs_fp64_grad_maxabs_rel"] = {
            tag: {k: float((g32[k].double() - g64[k]).abs().max() / (g64[k].abs().max() + 1e-300)) for k in g32}
            for tag, g32, g64 in (("d", d_grads, dg64), ("g", g_grads, gg64))}
    return out


def threshold_cases():
    """fp32 threshold semantics: torch's `tensor > python_float` compares in fp32."""
    cases = {}
    for thr in (0.02, 0.027, 0.02794):
        t32 = np.float32(thr)
        vals = np.array([np.nextafter(t32, np.float32(-1)), t32, np.nextafter(t32, np.float32(1)),
                         np.float32(0), np.float32(1)], dtype=np.float32)
        got = (torch.from_numpy(vals) > thr).to(torch.uint8).numpy()
        got64 = (vals.astype(np.float64) > thr).astype(np.uint8)
        cases[str(thr)] = {"vals_bits": vals.view(np.uint32).tolist(), "torch_gt": got.tolist(),
                           "numpy_f64_gt": got64.tolist()}
    return cases


def scene_case(gs):
    """Small scene (not a multiple of the patch) through the reference Generator with the
    tiling of evaluate.py:84-123 / data_loader.py:83-95 (crop 0), eval mode."""
    cfg = TINY
    rng = np.random.default_rng(99)
    h, w = 75, 52
    scene = rng.integers(1, 65536, size=(h, w, cfg["bands"]), dtype=np.uint16)
    scene[:9, :, :] = 0           # nodata border
    scene[:, :40, :][40:, :, :] = 0   # a fully invalid patch (rows 40.., cols ..40 region)
    G = Generator(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
    G.load_state_dict(gs)
    G.eval()
    patch = cfg["size"]
    pad_h = (0, patch - h % patch)
    pad_w = (0, patch - w % patch)
    padded = np.pad(scene, (pad_h, pad_w, (0, 0)), "reflect")
    diff = np.zeros((h, w))
    origins = [(r, c) for r in range(0, h, patch) for c in range(0, w, patch)]
    bs = 2
    for s in range(0, len(origins), bs):
        chunk = origins[s:s + bs]
        tiles = np.stack([padded[r:r + patch, c:c + patch] for r, c in chunk])
        x = torch.from_numpy(O.normalize_u16(tiles)).permute(0, 3, 1, 2).contiguous()
        if bool((x == -1).all()):
            continue
        with torch.no_grad():
            xf = G(x, torch.zeros(x.shape[0], 1))
            delta = (torch.abs(xf - x) / 2).numpy()
        delta = np.mean(delta, axis=1)
        for i, (r, c) in enumerate(chunk):
            hh = min(h - r, patch); ww = min(w - c, patch)
            diff[r:r + hh, c:c + ww] += delta[i, :hh, :ww]
    ora = O.predict_scene(gs, scene, patch, bs, False)
    assert np.abs(ora - diff).max() < 1e-6, "oracle predict_scene != reference tiling"
    return {"scene": scene, "diff": diff, "mask_0.02": (diff > 0.02).astype(np.uint8)}


VARIANTS = {
    # name: (act_g, act_d, n_feat_layers, interm_non_act, use_feats)
    "feats": ("relu", "lrelu", 2, False, True),
    "non_act": ("relu", "lrelu", 2, True, True),
    "non_act_allfeat": ("relu", "lrelu", 4, True, True),      # n_feat >= repeat: `main` is Identity
    "relu_d": ("lrelu", "relu", 2, False, False),
    "silu_g": ("silu", "lrelu", 4, False, False),
    "mish_g": ("mish", "lrelu", 4, False, True),
    "silu_d": ("relu", "silu", 2, False, True),
    "mish_d": ("silu", "mish", 2, True, True),
}


def ref_key(k, interm_non_act):
    """Oracle / default-layout key -> the reference module's key: with interm_non_act every layer is
    Sequential(act, conv), so the convolution sits at index 1 (fixed_point_gan.py:109-111)."""
    if interm_non_act:
        return re.sub(r"^(features|main)\.(\d+)\.0\.", r"\1.\2.1.", k)
    return k


def run_variants(cfg):
    """Optional configurations of the same step on the tiny config: the
    reference modules built with the variant's constructor arguments; losses and gradients (under
    the reference's own parameter names) stored; the oracle is checked against them on the way.
    Inputs are synth_batch / synth_*_state of the tiny config (numpy-seeded, regenerated by the tests)."""
    gs = O.synth_generator_state(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
    batch, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])
    t = {}
    worst = 0.0
    for name, (act_g, act_d, n_feat, ina, feats) in VARIANTS.items():
        ds = O.synth_discriminator_state(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"],
                                         n_feature_layers=n_feat)
        rds = {ref_key(k, ina): v for k, v in ds.items()}
        mk = lambda: ref_modules(gs, rds, cfg, act_g, act_d, n_feat, ina)   # noqa: E731
        G, D = mk()
        with torch.no_grad():
            r_src, r_cls, r_feat = D(batch[0])
        # (the reference returns the feature tensor as later in-place activations left it)
        t[f"{name}.D_feat"] = r_feat[0].numpy().copy()
        t[f"{name}.D_src"] = r_src.numpy()
        G, D = mk()
        _, dl, dg = ref_d_step(G, D, batch, alpha)
        G, D = mk()
        _, gl, gg = ref_g_step(G, D, batch, use_feats=feats)
        _, odl, odg = O.d_step(gs, ds, batch, alpha, act_g=act_g, act_d=act_d, new_stats={}, interm_non_act=ina)
        _, ogl, ogg = O.g_step(gs, ds, batch, act_g=act_g, act_d=act_d, new_stats={}, use_feats=feats,
                               interm_non_act=ina)
        with torch.no_grad():
            o_src, _, o_feat = O.discriminator_forward(ds, batch[0], act_d, ina)
        worst = max(worst, check(name + " D feat", o_feat[0], r_feat[0], 2e-6), check(name + " D src", o_src, r_src, 2e-6))
        for k in dl:
            worst = max(worst, check(f"{name} {k}", odl[k], dl[k], 1e-5))
            t[f"{name}.dloss.{k}"] = dl[k].numpy()
        for k in gl:
            worst = max(worst, check(f"{name} {k}", ogl[k], gl[k], 1e-5))
            t[f"{name}.gloss.{k}"] = gl[k].numpy()
        for k in odg:
            worst = max(worst, check(f"{name} dD/{k}", odg[k], dg[ref_key(k, ina)], 2e-3))
            t[f"{name}.dgrad.{ref_key(k, ina)}"] = dg[ref_key(k, ina)].numpy()
        for k in gg:
            worst = max(worst, check(f"{name} dG/{k}", ogg[k], gg[k], 2e-3))
            t[f"{name}.ggrad.{k}"] = gg[k].numpy()
        print(f"variant {name}: oracle vs reference worst rel-err so far {worst:.2e}")
    return t


TINY = dict(batch=2, size=32, bands=10, c_dim=1, g_conv_dim=8, d_conv_dim=8, g_repeat=2, d_repeat=3)
SMALL = dict(batch=3, size=64, bands=10, c_dim=1, g_conv_dim=16, d_conv_dim=16, g_repeat=3, d_repeat=5)
FULL = dict(batch=16, size=128, bands=10, c_dim=1, g_conv_dim=64, d_conv_dim=64, g_repeat=6, d_repeat=6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the full-size baseline config (scalars only)")
    ap.add_argument("--variants", action="store_true", help="only tests/golden/tiny_variants.npz")
    ap.add_argument("--full-only", action="store_true", help="only the full-size fixtures (full_scalars.json, full_grad_samples.npz)")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if args.variants:
        np.savez_compressed(os.path.join(GOLD, "tiny_variants.npz"), **run_variants(TINY))
        return

    if args.full_only:
        torch.set_num_threads(os.cpu_count())
        r = run_config(FULL, False, nsamp=4096)
        json.dump(r["scalars"], open(os.path.join(GOLD, "full_scalars.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(GOLD, "full_grad_samples.npz"), **r["big_samples"])
        print("full: worst oracle-vs-reference rel-err", r["scalars"]["oracle_vs_reference_worst_relerr"])
        return

    r = run_config(TINY, True)
    np.savez_compressed(os.path.join(GOLD, "tiny_step.npz"), **r["tensors"])
    json.dump(r["scalars"], open(os.path.join(GOLD, "tiny_scalars.json"), "w"), indent=1)
    print("tiny: worst oracle-vs-reference rel-err", r["scalars"]["oracle_vs_reference_worst_relerr"])

    r = run_config(SMALL, False)
    json.dump(r["scalars"], open(os.path.join(GOLD, "small_scalars.json"), "w"), indent=1)
    print("small: worst oracle-vs-reference rel-err", r["scalars"]["oracle_vs_reference_worst_relerr"])

    json.dump(threshold_cases(), open(os.path.join(GOLD, "threshold_cases.json"), "w"), indent=1)

    gs = O.synth_generator_state(TINY["g_conv_dim"], TINY["c_dim"], TINY["g_repeat"], TINY["bands"])
    sc = scene_case(gs)
    np.savez_compressed(os.path.join(GOLD, "tiny_scene.npz"), **sc)

    if args.full:
        torch.set_num_threads(os.cpu_count())
        r = run_config(FULL, False, nsamp=4096)
        json.dump(r["scalars"], open(os.path.join(GOLD, "full_scalars.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(GOLD, "full_grad_samples.npz"), **r["big_samples"])
        print("full: worst oracle-vs-reference rel-err", r["scalars"]["oracle_vs_reference_worst_relerr"])


if __name__ == "__main__":
    main()
