"""CPU oracle for the FCD Fixed-Point GAN hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``fcd_amd/`` imports it, and the HIP path raises when
its extension is missing instead of falling back to this code.

It is a *functional* restatement (state-dict in, tensors out; plain ``torch``
CPU ops in fp32 or fp64) of the reference algorithm:

* Generator / ResidualBlock   reference ``models/fixed_point_gan.py:22-84``
* Discriminator               reference ``models/fixed_point_gan.py:100-144``
* gradient penalty            reference ``fcd_solver.py:227-239``
* classification loss         reference ``fcd_solver.py:260-263``
* D-step loss graph           reference ``fcd_solver.py:279-308``
* G-step loss graph           reference ``fcd_solver.py:310-425``
* difference map / masks      reference ``fcd_solver.py:802-807, 762, 771-772``
* binarize / predict_scene    reference ``fcd_solver.py:566-567``,
                              ``evaluate.py:84-123``, ``data_loader.py:74-103``
* Adam update                 torch.optim.Adam as configured at
                              ``fcd_solver.py:181-187`` (lr 1e-4, betas (0.5, 0.999))

Pinning: ``oracle/make_golden.py`` (run in the build container, where
``/root/reference`` exists) imports the reference's own ``Generator`` and
``Discriminator`` classes, checks this restatement against them and writes the
fixtures under ``tests/golden/``.  ``fcd_solver.py`` itself cannot be imported
(Lightning, torchmetrics, cv2 ... are absent), so the loss graphs are pinned as
"reference modules + this file's restated loss arithmetic"; the reference has
no tests or golden vectors of its own.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]

IN_EPS = 1e-5          # nn.InstanceNorm2d default (fixed_point_gan.py:28,46)
IN_MOMENTUM = 0.1      # nn.InstanceNorm2d default
LRELU_SLOPE = 0.01     # fixed_point_gan.py:5-6


# --------------------------------------------------------------------------- #
# activations
# --------------------------------------------------------------------------- #
def _act(name: str, t: Tensor) -> Tensor:
    if name == "relu":
        return F.relu(t)
    if name == "lrelu":
        return F.leaky_relu(t, LRELU_SLOPE)
    if name == "silu":
        return F.silu(t)
    if name == "mish":
        return F.mish(t)
    raise NotImplementedError(name)


# --------------------------------------------------------------------------- #
# instance norm with optional tracked statistics
# --------------------------------------------------------------------------- #
def _instance_norm(t: Tensor, sd: StateDict, prefix: str, training: bool,
                   new_stats: Optional[StateDict]) -> Tensor:
    """InstanceNorm2d(affine=True[, track_running_stats=True]).

    Plain layers (no ``running_mean`` key) always use per-instance statistics.
    Tracked layers use per-instance statistics in training and update the
    running buffers with momentum 0.1 from the batch mean of per-instance means
    and of per-instance *unbiased* variances; in eval they normalise with the
    running buffers (BatchNorm-eval style).  ``num_batches_tracked`` is not
    touched by InstanceNorm.  (probed.)
    """
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    tracked = (prefix + ".running_mean") in sd
    if tracked and not training:
        rm = sd[prefix + ".running_mean"].to(t.dtype)
        rv = sd[prefix + ".running_var"].to(t.dtype)
        scale = w.to(t.dtype) / torch.sqrt(rv + IN_EPS)
        shift = b.to(t.dtype) - rm * scale
        return t * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    mean = t.mean(dim=(2, 3), keepdim=True)
    var = t.var(dim=(2, 3), unbiased=False, keepdim=True)
    if tracked and new_stats is not None:
        n = t.shape[2] * t.shape[3]
        src = new_stats if (prefix + ".running_mean") in new_stats else sd
        rm = src[prefix + ".running_mean"]
        rv = src[prefix + ".running_var"]
        bm = mean.detach().mean(dim=0).flatten()
        bv = (var.detach() * (n / (n - 1))).mean(dim=0).flatten()
        new_stats[prefix + ".running_mean"] = (1 - IN_MOMENTUM) * rm + IN_MOMENTUM * bm.to(rm.dtype)
        new_stats[prefix + ".running_var"] = (1 - IN_MOMENTUM) * rv + IN_MOMENTUM * bv.to(rv.dtype)
    xhat = (t - mean) / torch.sqrt(var + IN_EPS)
    return xhat * w.view(1, -1, 1, 1).to(t.dtype) + b.view(1, -1, 1, 1).to(t.dtype)


# --------------------------------------------------------------------------- #
# Generator
# --------------------------------------------------------------------------- #
def generator_layout(sd: StateDict) -> Tuple[int, int]:
    """(repeat_num, index of the first up-sampling layer) from state-dict keys."""
    rep = 0
    while f"main.{9 + rep}.main.0.weight" in sd:
        rep += 1
    return rep, 9 + rep


def generator_forward(sd: StateDict, x: Tensor, c: Tensor, training: bool = True,
                      activation: str = "relu",
                      new_stats: Optional[StateDict] = None) -> Tensor:
    """x_fake = G(x, c).  Reference ``models/fixed_point_gan.py:71-84``.

    ``new_stats``: if a dict is given (training only) the updated running
    buffers of the tracked norms are written into it, chaining across calls.
    """
    rep, up0 = generator_layout(sd)
    cmap = c.view(c.size(0), c.size(1), 1, 1).expand(-1, -1, x.size(2), x.size(3))
    h = torch.cat([x, cmap.to(x.dtype)], dim=1)
    h = F.conv2d(h, sd["main.0.weight"], None, 1, 3)
    h = _act(activation, _instance_norm(h, sd, "main.1", training, new_stats))
    for i in (3, 6):
        h = F.conv2d(h, sd[f"main.{i}.weight"], None, 2, 1)
        h = _act(activation, _instance_norm(h, sd, f"main.{i + 1}", training, new_stats))
    for r in range(rep):
        p = f"main.{9 + r}.main"
        t = F.conv2d(h, sd[p + ".0.weight"], None, 1, 1)
        t = _act(activation, _instance_norm(t, sd, p + ".1", training, new_stats))
        t = F.conv2d(t, sd[p + ".3.weight"], None, 1, 1)
        t = _instance_norm(t, sd, p + ".4", training, new_stats)
        h = h + t
    for i in (up0, up0 + 3):
        h = F.conv_transpose2d(h, sd[f"main.{i}.weight"], None, 2, 1)
        h = _act(activation, _instance_norm(h, sd, f"main.{i + 1}", training, new_stats))
    delta = F.conv2d(h, sd[f"main.{up0 + 6}.weight"], None, 1, 3)
    x_fake = torch.tanh(delta + x)
    invalid = (x == -1).to(x.dtype)
    return x_fake * (1 - invalid) - invalid


# --------------------------------------------------------------------------- #
# Discriminator
# --------------------------------------------------------------------------- #
def discriminator_convs(sd: StateDict) -> List[str]:
    """Prefixes of the strided conv layers in forward order."""
    names = []
    i = 0
    while f"features.{i}.0.weight" in sd:
        names.append(f"features.{i}.0")
        i += 1
    j = 0
    while f"main.{j}.0.weight" in sd:
        names.append(f"main.{j}.0")
        j += 1
    return names, i


def discriminator_forward(sd: StateDict, x: Tensor, activation: str = "lrelu", interm_non_act: bool = False):
    """(out_src, out_cls, [feat]) = D(x).  Reference ``fixed_point_gan.py:137-144``.

    ``interm_non_act`` (``:108-111``) builds each layer as ``act(inplace=True) -> conv``.  The
    in-place activation overwrites the previous layer's output -- including the tensor already
    appended to the feature list -- so every intermediate ends up activated as in the default
    network; only the last convolution's output stays un-activated (it feeds the heads, and it is
    the returned feature when ``n_feature_layers >= repeat_num``).  Pinned against the reference
    module in ``make_golden.py --variants``."""
    names, n_feat = discriminator_convs(sd)
    h = x
    feat = None
    for k, p in enumerate(names):
        h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], 2, 1)
        if not (interm_non_act and k == len(names) - 1):
            h = _act(activation, h)
        if k == n_feat - 1:
            feat = h
    out_src = F.conv2d(h, sd["conv1.weight"], None, 1, 1)
    out_cls = F.conv2d(h, sd["conv2.weight"], None, 1, 0)
    return out_src, out_cls.view(out_cls.size(0), out_cls.size(1)), [feat]


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
def gradient_penalty(y: Tensor, x: Tensor) -> Tensor:
    """mean((||dy/dx||_2 - 1)^2).  Reference ``fcd_solver.py:227-239``."""
    dydx = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=torch.ones_like(y),
                               retain_graph=True, create_graph=True, only_inputs=True)[0]
    dydx = dydx.reshape(dydx.size(0), -1)
    return torch.mean((torch.sqrt(torch.sum(dydx ** 2, dim=1)) - 1) ** 2)


def classification_loss(logit: Tensor, target: Tensor) -> Tensor:
    """Reference ``fcd_solver.py:260-263``."""
    return F.binary_cross_entropy_with_logits(logit, target.to(logit.dtype))


DEFAULT_LAMBDAS = dict(cls=1.0, rec=10.0, gp=10.0, id=10.0)   # fcd_main.py:86-89


def _leaf_params(sd: StateDict, grad: bool) -> StateDict:
    out = {}
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            out[k] = v.detach().clone().requires_grad_(grad)
        else:
            out[k] = v.detach().clone()
    return out


def d_step(g_sd: StateDict, d_sd: StateDict, batch, alpha: Tensor,
           lambdas=DEFAULT_LAMBDAS, act_g: str = "relu", act_d: str = "lrelu",
           new_stats: Optional[StateDict] = None, interm_non_act: bool = False):
    """Discriminator loss graph, ``fcd_solver.py:279-308`` with ``alpha`` injected.

    Lightning's ``toggle_optimizer`` freezes G for this step, so only D leaves
    carry gradients.  Returns (loss, losses dict, grads dict for D).
    """
    x_real, c_org, c_trg, label_org, label_trg = batch
    d = _leaf_params(d_sd, True)
    ina = interm_non_act
    out_src, out_cls, _ = discriminator_forward(d, x_real, act_d, ina)
    d_loss_real = -torch.mean(out_src)
    d_loss_cls = classification_loss(out_cls, label_org)
    with torch.no_grad():
        x_fake = generator_forward(g_sd, x_real, c_trg, True, act_g, new_stats)
    out_src, _, _ = discriminator_forward(d, x_fake.detach(), act_d, ina)
    d_loss_fake = torch.mean(out_src)
    alpha = alpha.to(x_real.dtype)
    x_hat = (alpha * x_real + (1 - alpha) * x_fake).detach().requires_grad_(True)
    out_src, _, _ = discriminator_forward(d, x_hat, act_d, ina)
    d_loss_gp = gradient_penalty(out_src, x_hat)
    loss = d_loss_real + d_loss_fake + lambdas["cls"] * d_loss_cls + lambdas["gp"] * d_loss_gp
    names = [k for k, v in d.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [d[k] for k in names], allow_unused=True)
    gd = {k: (g if g is not None else torch.zeros_like(d[k])) for k, g in zip(names, grads)}
    losses = {"D/loss_real": d_loss_real, "D/loss_fake": d_loss_fake,
              "D/loss_cls": d_loss_cls, "D/loss_gp": d_loss_gp}
    return loss.detach(), {k: v.detach() for k, v in losses.items()}, gd


def g_step(g_sd: StateDict, d_sd: StateDict, batch, lambdas=DEFAULT_LAMBDAS,
           act_g: str = "relu", act_d: str = "lrelu",
           new_stats: Optional[StateDict] = None, use_feats: bool = False, lambda_feat: float = 1.0,
           interm_non_act: bool = False):
    """Generator loss graph, ``fcd_solver.py:316-359, 404-406``; ``use_feats`` adds the
    feature-matching terms of ``:361-385`` (use_vgg off, use_mask hard-wired False at :341).
    D is frozen (Lightning toggle)."""
    ina = interm_non_act

    def feat_l1(fa, fb):
        return torch.mean(torch.stack([torch.mean(torch.abs(a - b)) for a, b in zip(fa, fb)]))

    x_real, c_org, c_trg, label_org, label_trg = batch
    g = _leaf_params(g_sd, True)
    if new_stats is None:
        new_stats = {}
    x_fake = generator_forward(g, x_real, c_trg, True, act_g, new_stats)
    out_src, out_cls, _ = discriminator_forward(d_sd, x_fake, act_d, ina)
    g_loss_fake = -torch.mean(out_src)
    g_loss_cls = classification_loss(out_cls, label_trg)

    x_fake_id = generator_forward(g, x_real, c_org, True, act_g, new_stats)
    out_src_id, out_cls_id, x_fake_id_feats = discriminator_forward(d_sd, x_fake_id, act_d, ina)
    g_loss_fake_id = -torch.mean(out_src_id)
    g_loss_cls_id = classification_loss(out_cls_id, label_org)
    g_loss_id = torch.mean(torch.abs(x_real - x_fake_id))

    x_reconst = generator_forward(g, x_fake, c_org, True, act_g, new_stats)
    g_loss_rec = torch.mean(torch.abs(x_real - x_reconst))

    x_reconst_id = generator_forward(g, x_fake_id, c_org, True, act_g, new_stats)
    g_loss_rec_id = torch.mean(torch.abs(x_real - x_reconst_id))

    g_loss_same = (g_loss_fake_id + lambdas["rec"] * g_loss_rec_id
                   + lambdas["cls"] * g_loss_cls_id + lambdas["id"] * g_loss_id)
    loss = g_loss_fake + lambdas["rec"] * g_loss_rec + lambdas["cls"] * g_loss_cls + g_loss_same
    losses = {"G/loss_fake": g_loss_fake, "G/loss_rec": g_loss_rec, "G/loss_cls": g_loss_cls,
              "G/loss_fake_id": g_loss_fake_id, "G/loss_rec_id": g_loss_rec_id,
              "G/loss_cls_id": g_loss_cls_id, "G/loss_id": g_loss_id}
    if use_feats:
        _, _, x_reconst_id_feats = discriminator_forward(d_sd, x_reconst_id, act_d, ina)
        _, _, x_real_feats = discriminator_forward(d_sd, x_real, act_d, ina)
        _, _, x_reconst_feats = discriminator_forward(d_sd, x_reconst, act_d, ina)
        g_loss_id_feat = feat_l1(x_real_feats, x_fake_id_feats)
        g_loss_rec_feat = feat_l1(x_real_feats, x_reconst_feats)
        g_loss_rec_id_feat = feat_l1(x_real_feats, x_reconst_id_feats)
        loss = loss + (lambda_feat * lambdas["rec"] * g_loss_rec_feat + lambda_feat * lambdas["rec"] * g_loss_rec_id_feat
                       + lambda_feat * lambdas["id"] * g_loss_id_feat)
        losses.update({"G/loss_id_feat": g_loss_id_feat, "G/loss_rec_feat": g_loss_rec_feat,
                       "G/loss_rec_id_feat": g_loss_rec_id_feat})
    names = [k for k, v in g.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [g[k] for k in names])
    return loss.detach(), {k: v.detach() for k, v in losses.items()}, dict(zip(names, grads))


# --------------------------------------------------------------------------- #
# Adam (torch.optim.Adam, no weight decay, no amsgrad)
# --------------------------------------------------------------------------- #
def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-4,
                beta1: float = 0.5, beta2: float = 0.999, eps: float = 1e-8):
    """One Adam update in torch's operation order; returns (p, m, v)."""
    m = m + (g - m) * (1 - beta1)                       # lerp
    v = v * beta2 + (g * g) * (1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - step_size * (m / denom), m, v


# --------------------------------------------------------------------------- #
# inference: difference map, masks, scene tiling
# --------------------------------------------------------------------------- #
def difference_map(g_sd: StateDict, x: Tensor, training: bool = False,
                   act_g: str = "relu") -> Tensor:
    """mean_c(|G(x, 0) - x| / 2).  Reference ``fcd_solver.py:802-807``."""
    c = torch.zeros(x.shape[0], 1, dtype=x.dtype)
    with torch.no_grad():
        x_fake = generator_forward(g_sd, x, c, training, act_g, None)
        return torch.mean(torch.abs(x_fake - x) / 2, dim=1)


def threshold_mask(diff: Tensor, thr: float) -> Tensor:
    """``(difference > thr).to(uint8)``: torch compares in fp32 with ``thr`` rounded to
    fp32 (``fcd_solver.py:762, 511``)."""
    return (diff > thr).to(torch.uint8)


def pseudo_mask(diff: Tensor, thr: float, valid: Tensor) -> Tensor:
    """``+1`` then invalid -> 0 (``fcd_solver.py:771-772``): 0 invalid, 1 clear, 2 cloudy."""
    m = threshold_mask(diff, thr) + 1
    m[~valid] = 0
    return m


def binarize(diff: np.ndarray, thr: float = 0.2) -> np.ndarray:
    """numpy ``>`` in the array's own dtype (float64 for a stitched scene).
    Reference ``fcd_solver.py:566-567``."""
    return (diff > thr).astype(np.uint8)


def compute_confusion_matrix(prediction: np.ndarray, ground_truth: np.ndarray, num_classes: int) -> np.ndarray:
    """``metrics.py:4-15``: ``cm[truth][prediction]`` counts."""
    idx = ground_truth.astype(np.int64).ravel() * num_classes + prediction.astype(np.int64).ravel()
    return np.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def f1_score(cm: np.ndarray) -> float:
    """``metrics.py:33-45`` (macro F1; an empty class scores 1)."""
    f1s = []
    for i in range(cm.shape[0]):
        tp = int(cm[i, i])
        fp = int(cm[:, i].sum()) - tp
        fn = int(cm[i, :].sum()) - tp
        denom = 2 * tp + fp + fn
        f1s.append(1 if denom == 0 else 2 * float(tp) / denom)
    return float(np.mean(f1s))


def find_best_threshold(all_preds: np.ndarray, all_targets: np.ndarray, n_thresholds: int = 30) -> float:
    """``fcd_solver.py:708-721`` on already gathered valid pixels (``all_preds`` fp32 difference
    values, ``all_targets`` in {0, 1}).  ``all_preds > threshold`` is evaluated in fp32 with the
    threshold rounded to fp32: numpy 1.x (value-based casting, the reference's era) does that for a
    float32 array against a float64 scalar; numpy 2 would compare in float64."""
    thresholds = np.linspace(start=all_preds.min(), stop=all_preds.max(), num=n_thresholds)
    best_f1, best_threshold = None, None
    for threshold in thresholds:
        preds = (all_preds > np.float32(threshold)).astype(np.uint8)
        f1 = f1_score(compute_confusion_matrix(preds, all_targets, 2))
        if best_f1 is None or f1 >= best_f1:
            best_f1, best_threshold = f1, float(threshold)
        else:
            break
    return best_threshold


def normalize_u16(v: np.ndarray) -> np.ndarray:
    """uint16 -> [-1, 1] fp32: ``(v / 65535 - 0.5) / 0.5`` (``data_loader.py:375``,
    albumentations Normalize(mean .5, std .5, max 65535)): fp32 arithmetic as
    ``(v - 32767.5) * fp32(1/32767.5)``; 0 maps to exactly -1.0."""
    t = v.astype(np.float32)
    t -= np.float32(32767.5)
    t *= np.reciprocal(np.float32(32767.5), dtype=np.float32)
    return t


def patch_batch(images: np.ndarray, index: np.ndarray, aug: Optional[np.ndarray] = None,
                masks: Optional[np.ndarray] = None):
    """Batch assembly of the train / validation loader: ``L8BiomeHDFDataset.__getitem__``
    (``data_loader.py:287-313``) through the transform chain of ``get_dataset`` (``:370-377``) and
    DataLoader's collate.  images (N,H,W,C) uint16, masks (N,H,W) uint8; aug[b] = (flags, dy, dx):
    HorizontalFlip (bit 0, ``img[:, ::-1]``), VerticalFlip (bit 1, ``img[::-1]``), then a whole-pixel
    translation whose uncovered pixels are 0; Normalize; HWC -> CHW.  Returns (x fp32 (B,C,H,W),
    masks uint8 (B,H,W) or None)."""
    xs, ms = [], []
    for b, i in enumerate(index):
        img = images[i]
        m = masks[i] if masks is not None else None
        flags, dy, dx = (0, 0, 0) if aug is None else (int(v) for v in aug[b])
        if flags & 1:
            img = img[:, ::-1]
            m = m[:, ::-1] if m is not None else None
        if flags & 2:
            img = img[::-1]
            m = m[::-1] if m is not None else None
        if dy or dx:
            h, w = img.shape[:2]
            sh = np.zeros_like(img)
            ys, yd = (slice(0, h - dy), slice(dy, h)) if dy >= 0 else (slice(-dy, h), slice(0, h + dy))
            xsrc, xdst = (slice(0, w - dx), slice(dx, w)) if dx >= 0 else (slice(-dx, w), slice(0, w + dx))
            sh[yd, xdst] = img[ys, xsrc]
            img = sh
            if m is not None:
                shm = np.zeros_like(m)
                shm[yd, xdst] = m[ys, xsrc]
                m = shm
        xs.append(np.transpose(normalize_u16(np.ascontiguousarray(img)), (2, 0, 1)))
        if m is not None:
            ms.append(np.ascontiguousarray(m))
    return np.stack(xs), (np.stack(ms) if masks is not None else None)


def scene_patches(h: int, w: int, patch: int, crop: int = 0):
    """Row-major (row, col) patch origins and reflect-pad amounts.
    Reference ``data_loader.py:83-95``."""
    cps = patch - 2 * crop
    pad_h = (crop, (cps - h % cps) + crop)
    pad_w = (crop, (cps - w % cps) + crop)
    origins = [(r, c) for r in range(0, h, cps) for c in range(0, w, cps)]
    return origins, pad_h, pad_w, cps


def predict_scene(g_sd: StateDict, scene_u16: np.ndarray, patch: int = 128, batch: int = 64,
                  training: bool = False, act_g: str = "relu") -> np.ndarray:
    """Tile, run G towards "clear", band-mean of |x_fake - x| / 2, stitch into float64.
    Reference ``evaluate.py:84-123`` (crop_size 0)."""
    assert scene_u16.dtype == np.uint16
    h, w, _ = scene_u16.shape
    origins, pad_h, pad_w, cps = scene_patches(h, w, patch, 0)
    padded = np.pad(scene_u16, (pad_h, pad_w, (0, 0)), "reflect")
    diff = np.zeros((h, w))
    for s in range(0, len(origins), batch):
        chunk = origins[s:s + batch]
        tiles = np.stack([padded[r:r + patch, c:c + patch] for r, c in chunk])
        x = torch.from_numpy(normalize_u16(tiles)).permute(0, 3, 1, 2).contiguous()
        if bool((x == -1).all()):
            continue
        cz = torch.zeros(x.shape[0], 1)
        with torch.no_grad():
            x_fake = generator_forward(g_sd, x, cz, training, act_g, None)
            delta = (torch.abs(x_fake - x) / 2).numpy()
        delta = np.mean(delta, axis=1)
        for i, (r, c) in enumerate(chunk):
            hh = min(h - r, cps)
            ww = min(w - c, cps)
            diff[r:r + hh, c:c + ww] += delta[i, :hh, :ww]
    return diff


# --------------------------------------------------------------------------- #
# synthetic inputs (bit-reproducible across machines: numpy PCG64 only)
# --------------------------------------------------------------------------- #
def synth_batch(b: int, size: int = 128, bands: int = 10, seed: int = 8888):
    """SURVEY.md section 8(d) recipe: uint16 patches -> [-1, 1], an all-bands
    nodata rectangle in every 4th sample, Bernoulli labels, permuted targets, alpha."""
    rng = np.random.default_rng(seed)
    raw = rng.integers(1, 65536, size=(b, size, size, bands), dtype=np.uint16)
    rh, rw = max(1, size // 8), max(1, size // 4)
    raw[::4, :rh, :rw, :] = 0
    x = torch.from_numpy(normalize_u16(raw)).permute(0, 3, 1, 2).contiguous()
    label_org = torch.from_numpy((rng.random((b, 1)) < 0.5).astype(np.float32))
    perm = rng.permutation(b)
    label_trg = label_org[torch.from_numpy(perm)]
    alpha = torch.from_numpy(rng.random((b, 1, 1, 1)).astype(np.float32))
    return (x, label_org.clone(), label_trg.clone(), label_org, label_trg), alpha


def _uniform(rng, shape, bound):
    return torch.from_numpy(((rng.random(shape) * 2 - 1) * bound).astype(np.float32))


def synth_generator_state(conv_dim=64, c_dim=1, repeat_num=6, num_channels=10, seed=1234) -> StateDict:
    """numpy-seeded default-init-like weights (uniform +-1/sqrt(fan_in)), IN gamma
    perturbed around 1 and beta around 0 so that affine terms are exercised."""
    rng = np.random.default_rng(seed)
    sd: StateDict = {}

    def conv(name, co, ci, k):
        sd[name] = _uniform(rng, (co, ci, k, k), 1 / math.sqrt(ci * k * k))

    def norm(name, ch, tracked):
        sd[name + ".weight"] = 1 + _uniform(rng, (ch,), 0.2)
        sd[name + ".bias"] = _uniform(rng, (ch,), 0.2)
        if tracked:
            sd[name + ".running_mean"] = _uniform(rng, (ch,), 0.1)
            sd[name + ".running_var"] = 1 + _uniform(rng, (ch,), 0.2)
            sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    conv("main.0.weight", conv_dim, num_channels + c_dim, 7)
    norm("main.1", conv_dim, False)
    d = conv_dim
    for i in (3, 6):
        conv(f"main.{i}.weight", d * 2, d, 4)
        norm(f"main.{i + 1}", d * 2, False)
        d *= 2
    for r in range(repeat_num):
        p = f"main.{9 + r}.main"
        conv(p + ".0.weight", d, d, 3)
        norm(p + ".1", d, True)
        conv(p + ".3.weight", d, d, 3)
        norm(p + ".4", d, True)
    up0 = 9 + repeat_num
    for i in (up0, up0 + 3):
        # ConvTranspose2d weight is [C_in, C_out, kh, kw]; torch's fan_in uses dim 1.
        sd[f"main.{i}.weight"] = _uniform(rng, (d, d // 2, 4, 4), 1 / math.sqrt((d // 2) * 16))
        norm(f"main.{i + 1}", d // 2, False)
        d //= 2
    conv(f"main.{up0 + 6}.weight", num_channels, d, 7)
    return sd


def synth_discriminator_state(image_size=128, conv_dim=64, c_dim=1, repeat_num=6, num_channels=10,
                              n_feature_layers=4, seed=4321) -> StateDict:
    rng = np.random.default_rng(seed)
    sd: StateDict = {}
    ci, co = num_channels, conv_dim
    for k in range(repeat_num):
        p = f"features.{k}.0" if k < n_feature_layers else f"main.{k - n_feature_layers}.0"
        bound = 1 / math.sqrt(ci * 16)
        sd[p + ".weight"] = _uniform(rng, (co, ci, 4, 4), bound)
        sd[p + ".bias"] = _uniform(rng, (co,), bound)
        ci, co = co, co * 2
    ks = image_size // (2 ** repeat_num)
    sd["conv1.weight"] = _uniform(rng, (1, ci, 3, 3), 1 / math.sqrt(ci * 9))
    sd["conv2.weight"] = _uniform(rng, (c_dim, ci, ks, ks), 1 / math.sqrt(ci * ks * ks))
    return sd


def to_dtype(sd: StateDict, dtype) -> StateDict:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
