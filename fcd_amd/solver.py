"""FCD solver on the HIP path: the reference ``FCDSolver`` API without Lightning.

Mirrors ``fcd_solver.py``: ``training_step(batch, batch_idx, optimizer_idx)`` (:272),
``gradient_penalty`` (:227), ``classification_loss`` (:260), ``configure_optimizers``
(:181), ``optimizer_step`` gating (:549-564), ``compute_difference_map`` (:802),
``make_psuedo_masks`` (:729, also exposed as ``generate_masks``) and ``binarize`` (:566).
What Lightning did around those hooks (toggle_optimizer, closure, backward,
zero_grad(set_to_none=True)) is ``train_step``.  Every tensor op on the path is a
launch into libfcd_hip.so; torch supplies memory, streams, autograd bookkeeping
and the (optional) RCCL process group.
"""
from __future__ import annotations

import gc
import os
from argparse import Namespace
from typing import Dict, Iterable, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import losses, ops
from .arena import ParamArena
from .discriminator import Discriminator, input_grad_only
from .generator import Generator


def default_config(**over) -> Namespace:
    """Hot-path defaults of ``fcd_main.py:79-158``."""
    cfg = dict(c_dim=1, image_size=128, g_conv_dim=64, d_conv_dim=64, g_repeat_num=6, d_repeat_num=6,
               lambda_cls=1.0, lambda_rec=10.0, lambda_gp=10.0, lambda_id=10.0, lambda_feat=1.0, lambda_vgg=1.0,
               dataset="L8Biome", batch_size=16, num_iters=200000, num_iters_decay=100000, g_lr=1e-4, d_lr=1e-4,
               n_critic=5, beta1=0.5, beta2=0.999, act_D="lrelu", act_G="relu", n_feat_layers=4,
               interm_non_act=False, use_attention=False, use_feats=False, use_vgg=False, init_type="none",
               num_channels=10, mode="train", precision="fp32", device="cuda")
    cfg.update(over)
    return Namespace(**cfg)


def f1_score(cm: np.ndarray) -> float:
    """Macro F1 of a confusion matrix ``cm[truth][prediction]`` (``metrics.py:33-45``)."""
    f1s = []
    for i in range(cm.shape[0]):
        tp = int(cm[i, i])
        fp = int(cm[:, i].sum()) - tp
        fn = int(cm[i, :].sum()) - tp
        denom = 2 * tp + fp + fn
        f1s.append(1 if denom == 0 else 2 * float(tp) / denom)
    return float(np.mean(f1s))


def iou_score(cm: np.ndarray) -> float:
    """Mean IoU of ``cm[truth][prediction]`` (``metrics.py:16-28``; an empty class scores 1)."""
    ious = []
    for i in range(cm.shape[0]):
        tp = int(cm[i, i])
        denom = int(cm[:, i].sum()) + int(cm[i, :].sum()) - tp
        ious.append(1 if denom == 0 else float(tp) / denom)
    return float(np.mean(ious))


def accuracy(cm: np.ndarray) -> float:
    """Overall accuracy (``metrics.py`` ``accuracy``): trace / total."""
    total = int(cm.sum())
    return float(np.trace(cm)) / total if total else 1.0


def best_threshold_from_hist(hist: np.ndarray, thresholds: np.ndarray) -> float:
    """The reference's sweep (``fcd_solver.py:713-721``: keep going while F1 does not drop) on the
    device histogram: ``hist[cls][j]`` = valid pixels of class ``cls`` predicted 1 for exactly the
    first ``j`` thresholds."""
    total = hist.sum(axis=1)
    best_f1, best_thr = None, None
    for t, thr in enumerate(thresholds):
        pred1 = hist[:, t + 1:].sum(axis=1)
        cm = np.array([[total[0] - pred1[0], pred1[0]], [total[1] - pred1[1], pred1[1]]], dtype=np.int64)
        f1 = f1_score(cm)
        if best_f1 is None or f1 >= best_f1:
            best_f1, best_thr = f1, float(thr)
        else:
            break
    return best_thr


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas) semantics (``fcd_solver.py:183,186``) with the HIP update.

    With an ``arena`` (:class:`fcd_amd.arena.ParamArena` over exactly these parameters) the
    whole network is updated by ONE launch over the flat buffers; that equals the
    per-parameter update as long as every parameter receives a gradient in every step it
    takes part in (true for the FCD losses: a parameter without gradient keeps m = v = 0)."""

    def __init__(self, params, lr=1e-4, betas=(0.5, 0.999), eps=1e-8, arena=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.arena = arena
        self._flat_state = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.arena is not None and self.arena.intact():
            self._step_arena()
            return loss
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"], st["step"])
                torch.autograd.graph.increment_version(p)    # raw-pointer update: invalidate packed copies
        return loss

    def load_state_dict(self, state_dict):
        """The loaded per-parameter moments are copied into the flat buffers at the next arena step."""
        super().load_state_dict(state_dict)
        self._flat_state = None

    def _step_arena(self):
        ar = self.arena
        if len(self.param_groups) != 1:
            raise RuntimeError("FusedAdam with an arena takes a single parameter group")
        group = self.param_groups[0]
        if self._flat_state is None:
            m, v = torch.zeros_like(ar.flat), torch.zeros_like(ar.flat)
            step = 0
            for p, off in zip(ar.params, ar.offsets):      # per-parameter views: the usual state_dict layout
                mv, vv = m[off:off + p.numel()].view(p.shape), v[off:off + p.numel()].view(p.shape)
                old = self.state.get(p)
                if old:                                     # state that load_state_dict (or per-parameter steps) left
                    mv.copy_(old["exp_avg"])
                    vv.copy_(old["exp_avg_sq"])
                    step = max(step, int(old["step"]))
                self.state[p] = {"step": step, "exp_avg": mv, "exp_avg_sq": vv}
            self._flat_state = {"step": step, "exp_avg": m, "exp_avg_sq": v}
        fs = self._flat_state
        fs["step"] += 1
        b1, b2 = group["betas"]
        ops.adam_step(ar.flat, ar.grad, fs["exp_avg"], fs["exp_avg_sq"], group["lr"], b1, b2, group["eps"], fs["step"])
        for p in ar.params:
            self.state[p]["step"] = fs["step"]
        ar.bump_versions()


class FCDSolver(nn.Module):
    """Solver for training and testing Fixed-Point GAN for Cloud Detection (HIP path)."""

    def __init__(self, config=None, **over):
        super().__init__()
        if config is None:
            config = default_config(**over)
        elif isinstance(config, dict):
            config = Namespace(**{**vars(default_config()), **config})
        self.config = config
        for k in ("c_dim", "image_size", "g_conv_dim", "d_conv_dim", "g_repeat_num", "d_repeat_num", "lambda_cls",
                  "lambda_rec", "lambda_gp", "lambda_id", "num_channels", "g_lr", "d_lr", "n_critic", "beta1", "beta2",
                  "dataset", "batch_size"):
            setattr(self, k, getattr(config, k))
        self.act_G, self.act_D = config.act_G, config.act_D
        self.n_feature_layers = config.n_feat_layers
        if getattr(config, "use_vgg", False):
            raise NotImplementedError("use_vgg (a torchvision VGG feature loss) is outside the HIP hot path "
                                      "(SURVEY section 2 row 10)")
        self.use_feats = bool(getattr(config, "use_feats", False))
        self.lambda_feat = float(getattr(config, "lambda_feat", 1.0))      # fcd_main.py:90
        self.interm_non_act = bool(getattr(config, "interm_non_act", False))
        self.threshold = 0.027                      # fcd_solver.py:79
        self.global_step = 0
        self.loss_dict: Dict[str, torch.Tensor] = {}
        self.alpha_override: Optional[torch.Tensor] = None   # inject alpha (fcd_solver.py:296) for parity tests
        self.grad_sync = None                       # parallel.GradSync when data-parallel
        self.overlap_comm = True                    # overlap D's gradient all-reduce with the G step's start
        self.bucket_overlap = True                  # send gradient spans to the all-reduce from inside backward
        self.bucket_mb = (64.0, 8.0)                # span sizes for (D: 179 MB of gradients, G: 33.8 MB)
        self._reducers = {}
        self._scheds = None
        self.pair_g_calls = True                    # G step: the four generator calls as two paired-batch calls
        self.pair_d_calls = True                    # D step: D(x_real) and D(x_fake) as one paired-batch call
        self._param_lists = {}                      # _set_requires_grad: id(module) -> (module, parameters)
        # train_step only (training_step called by hand keeps the reference's own sequence):
        self.share_generator_forward = True         # the D step's G(x_real, c_trg) IS the G step's first generator call
        self.skip_unused_d_forward = True           # D(x_reconst_id) (reference :358) feeds nothing unless use_feats
        self._shared_fake = None
        self._in_train_step = False
        self._pending = None
        self._val_cm = None
        self.freeze_host_objects = True             # gc.freeze() after the first train_step (see there)
        self.overflow_check_every = 64              # fp32x2: poll the overflow word every so many train steps
        # fp32x2 / fp16: the discriminator on the scaled-fp16 kernels too (maxima handed from layer to layer through
        # FcdConvDesc.out_amax).  Parity-tested, but measured level with BF16X3 / bf16 in both modes (DESIGN.md section 6d:
        # its weight-bound layers gain nothing), so it is off by default; FCD_D_H2=1 for A/B runs
        self.d_scaled_fp16 = os.environ.get("FCD_D_H2", "0") == "1"
        self._host_frozen = False
        self.build_model()
        self.D.register_forward_pre_hook(self._flush_pending)
        self.set_precision(getattr(config, "precision", "fp32"))
        self._opts = None
        self.arenas = (None, None)

    # ---------------------------------------------------------------- model -------------
    def build_model(self):
        """Create a generator and a discriminator (``fcd_solver.py:150-164``, init_type 'none')."""
        self.G = Generator(self.g_conv_dim, self.c_dim, self.g_repeat_num, self.num_channels, activation=self.act_G)
        self.D = Discriminator(self.image_size, self.d_conv_dim, self.c_dim, self.d_repeat_num, self.num_channels,
                               activation=self.act_D, n_feature_layers=self.n_feature_layers,
                               interm_non_act=self.interm_non_act)

    def set_precision(self, precision: str):
        """'fp32': exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere.
        'fp32x3': fp32-faithful results on the bf16 pipe: every fp32 operand is split in registers
        into three bf16 pieces (exactly, up to 2^-24), six bf16 MFMAs per product group, fp32
        accumulate; measured error against fp64 no larger than the exact-fp32 MFMA's, convolutions
        20-40 % faster.  The 10/12-channel image layers keep their exact-fp32 16-wide kernels.
        'fp32x2': as fp32x3, but the generator's convolutions that take pre-split weights (and the weight gradient of
        its 3x3 layers) split each operand, scaled by a power of two taken from the tensor's max |.|, into two fp16
        pieces: three MFMAs per product group instead of six, same parity bars.  The maxima come from the kernels
        that write the operands (InstanceNorm apply / backward).  The discriminator runs as under fp32x3.
        'fp16': the reduced-precision variant ("16-bit MFMA" of BASELINE.json configs[1]): fp32x2's kernels, operands and
        maxima in the generator, but only the high fp16 piece of every operand (11 significant bits; bf16 has 8) and ONE
        MFMA per product group; the discriminator with operands rounded to bf16.  fp32 accumulate, fp32 tensors in HBM.
        'bf16': operands rounded to bf16 at the MFMA fragment everywhere, fp32 accumulate."""
        p, pw = {"fp32": (L.PREC_F32, None), "fp32x3": (L.PREC_BF16X3, None), "bf16": (L.PREC_BF16, None),
                 "fp32x2": (L.PREC_F16X2, None), "fp16": (L.PREC_F16, None)}[precision]
        for net in (self.G, self.D):
            if net is self.D and p in L.F16_FAMILY and (self.D.smooth or not self.d_scaled_fp16):
                # SiLU / Mish discriminators keep their un-fused activation launches and no maxima bookkeeping: split bf16
                # pieces under fp32x2, plain bf16 operands under the reduced-precision variant
                net.precision = L.PREC_BF16X3 if p == L.PREC_F16X2 else L.PREC_BF16
            else:
                net.precision = p
            net.wgrad_precision = pw
            for m in net.modules():          # drop packed weights of the other precision's layout
                if hasattr(m, "_packed"):
                    m._packed = {}
        self.precision = precision

    @property
    def device(self):
        return next(self.G.parameters()).device

    def configure_optimizers(self, arenas: bool = False):
        """``arenas``: move each network's parameters / gradients into flat arenas (see
        :mod:`fcd_amd.arena`); do it after the final ``.to(device)``."""
        ar_d = ParamArena(self.D.parameters()) if arenas else None
        ar_g = ParamArena(self.G.parameters()) if arenas else None
        self.arenas = (ar_d, ar_g)
        opt_D = FusedAdam(self.D.parameters(), self.d_lr, (self.beta1, self.beta2), arena=ar_d)
        sched_D = torch.optim.lr_scheduler.ReduceLROnPlateau(opt_D, mode="max", patience=10)
        opt_G = FusedAdam(self.G.parameters(), self.g_lr, (self.beta1, self.beta2), arena=ar_g)
        sched_G = torch.optim.lr_scheduler.ReduceLROnPlateau(opt_G, mode="max", patience=10)
        return ({"optimizer": opt_D, "lr_scheduler": {"scheduler": sched_D, "monitor": "val/F1Score"}},
                {"optimizer": opt_G, "lr_scheduler": {"scheduler": sched_G, "monitor": "val/F1Score"}})

    def forward(self, x_real, c_org, c_trg, label_org, label_trg):
        x_fake = self.G(x_real, c_trg)
        out_src, out_cls, _ = self.D(x_fake)
        return out_src, out_cls

    # ---------------------------------------------------------------- losses ------------
    def gradient_penalty(self, y, x):
        """Compute gradient penalty: (L2_norm(dy/dx) - 1)**2."""
        weight = torch.ones(y.size(), device=y.device)
        with input_grad_only(self.D):
            dydx = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=weight, retain_graph=True,
                                       create_graph=True, only_inputs=True)[0]
        return losses.gp_loss(dydx.view(dydx.size(0), -1))

    def classification_loss(self, logit, target, dataset="L8Biome"):
        """Compute binary cross entropy loss."""
        if dataset in ["L8Biome"]:
            return losses.bce_logits_mean(logit, target)
        raise NotImplementedError(dataset)

    @staticmethod
    def _pair_fits(x, widest: int) -> bool:
        """A paired batch doubles every activation.  The library's kernels address a window of a few images
        with 32-bit lane offsets (tensors themselves may exceed 2 GiB), so what has to fit is two images of
        the largest activation (full resolution, `widest` channels, fp32)."""
        _, _, h, w = x.shape
        return 2 * h * w * max(int(widest), 12) * 4 < (1 << 31)

    @staticmethod
    def _feat_l1(real_feats, other_feats):
        terms = [losses.l1_mean(a.detach(), b) for a, b in zip(real_feats, other_feats)]
        return terms[0] if len(terms) == 1 else torch.mean(torch.stack(terms))

    def _alpha(self, b, device):
        if self.alpha_override is not None:
            return self.alpha_override.to(device).contiguous().float()
        # the reference draws alpha from the CPU generator (fcd_solver.py:296) and copies it over; a pageable
        # copy stops the host until the stream has drained, so the draw goes through pinned memory instead
        a = torch.rand(b, 1, 1, 1)
        return a.pin_memory().to(device, non_blocking=True) if device.type == "cuda" else a.to(device)

    def training_step(self, batch, batch_idx, optimizer_idx):
        x_real, c_org, c_trg, label_org, label_trg = batch
        loss_dict = {}
        loss = None
        if optimizer_idx == 0:
            # ---- train the discriminator (fcd_solver.py:279-308)
            if self.pair_d_calls and self._pair_fits(x_real, self.d_conv_dim // 4):
                # D(x_real) and D(x_fake) as one call on the paired batch (D has no cross-sample coupling)
                nb = x_real.size(0)
                x_fake = self._shared_fake[:nb] if self._shared_fake is not None else self.G(x_real, c_trg)
                src2, cls2, _ = self.D(torch.cat([x_real, x_fake.detach()]))
                d_loss_real = losses.mean_scaled(src2[:nb], -1.0)
                d_loss_cls = self.classification_loss(cls2[:nb], label_org, self.dataset)
                d_loss_fake = losses.mean_scaled(src2[nb:], 1.0)
            else:
                out_src, out_cls, _ = self.D(x_real)
                d_loss_real = losses.mean_scaled(out_src, -1.0)
                d_loss_cls = self.classification_loss(out_cls, label_org, self.dataset)

                x_fake = self._shared_fake[:x_real.size(0)] if self._shared_fake is not None else self.G(x_real, c_trg)
                out_src, out_cls, _ = self.D(x_fake.detach())
                d_loss_fake = losses.mean_scaled(out_src, 1.0)

            alpha = self._alpha(x_real.size(0), x_real.device)
            x_hat = ops.lerp_rows(x_real.detach().contiguous(), x_fake.detach().contiguous(),
                                  alpha.view(-1)).requires_grad_(True)
            out_src, _, _ = self.D(x_hat)
            d_loss_gp = self.gradient_penalty(out_src, x_hat)

            loss = d_loss_real + d_loss_fake + self.lambda_cls * d_loss_cls + self.lambda_gp * d_loss_gp
            loss_dict["D/loss_real"] = d_loss_real
            loss_dict["D/loss_fake"] = d_loss_fake
            loss_dict["D/loss_cls"] = d_loss_cls
            loss_dict["D/loss_gp"] = d_loss_gp
        elif optimizer_idx == 1:
            # ---- train the generator (fcd_solver.py:310-425)
            if (self.global_step + 1) % self.n_critic == 0:
                if self.pair_g_calls and self._pair_fits(x_real, self.g_conv_dim):
                    # The reference's four generator calls as two calls on paired batches -- (x_real, c_trg)
                    # with (x_real, c_org), then (x_fake, c_org) with (x_fake_id, c_org) -- and D once on
                    # the first pair: per-sample results are those of the separate calls (InstanceNorm and
                    # every loss term are per sample; Generator.sample_groups keeps the running-statistics
                    # updates in the reference's order), and every launch-bound kernel runs half as often.
                    nb = x_real.size(0)
                    if self._shared_fake is not None:
                        fake2, self._shared_fake = self._shared_fake, None      # made by train_step before the D step
                    else:
                        with self.G.sample_groups(2):
                            fake2 = self.G(torch.cat([x_real, x_real]), torch.cat([c_trg, c_org]))
                    x_fake, x_fake_id = fake2[:nb], fake2[nb:]
                    # (the reconstruction call first: it does not need D, so a deferred D update -- data parallel: D's
                    #  gradient all-reduce -- keeps running beside it; D's forward pre-hook below waits for it)
                    with self.G.sample_groups(2):
                        rec2 = self.G(fake2, torch.cat([c_org, c_org]))
                    x_reconst, x_reconst_id = rec2[:nb], rec2[nb:]
                    src2, cls2, feats2 = self.D(fake2)
                    out_src, out_cls = src2[:nb], cls2[:nb]
                    out_src_id, out_cls_id, x_fake_id_feats = src2[nb:], cls2[nb:], [f[nb:] for f in feats2]
                else:
                    x_fake = self.G(x_real, c_trg)
                    out_src, out_cls, _ = self.D(x_fake)
                    x_fake_id = self.G(x_real, c_org)
                    out_src_id, out_cls_id, x_fake_id_feats = self.D(x_fake_id)
                    x_reconst = self.G(x_fake, c_org)
                    x_reconst_id = self.G(x_fake_id, c_org)
                g_loss_fake = losses.mean_scaled(out_src, -1.0)
                g_loss_cls = self.classification_loss(out_cls, label_trg, self.dataset)

                g_loss_fake_id = losses.mean_scaled(out_src_id, -1.0)
                g_loss_cls_id = self.classification_loss(out_cls_id, label_org, self.dataset)
                g_loss_id = losses.l1_mean(x_real, x_fake_id)

                g_loss_rec = losses.l1_mean(x_real, x_reconst)

                if self.use_feats or not self._in_train_step or not self.skip_unused_d_forward:
                    _, _, x_reconst_id_feats = self.D(x_reconst_id)    # reference :358
                # (else: the reference computes this forward and drops its outputs -- only use_feats reads them, D has
                #  no buffers, nothing requires its graph: a dead call, left out by train_step)
                g_loss_rec_id = losses.l1_mean(x_real, x_reconst_id)

                if self.use_feats:
                    # feature matching on D's exposed feature map (fcd_solver.py:361-385; the list has
                    # one entry, so mean(stack([m])) == m)
                    _, _, x_real_feats = self.D(x_real)
                    g_loss_id_feat = self._feat_l1(x_real_feats, x_fake_id_feats)
                    _, _, x_reconst_feats = self.D(x_reconst)
                    g_loss_rec_feat = self._feat_l1(x_real_feats, x_reconst_feats)
                    g_loss_rec_id_feat = self._feat_l1(x_real_feats, x_reconst_id_feats)
                    g_loss_feat = (self.lambda_feat * self.lambda_rec * g_loss_rec_feat
                                   + self.lambda_feat * self.lambda_rec * g_loss_rec_id_feat
                                   + self.lambda_feat * self.lambda_id * g_loss_id_feat)

                g_loss_same = (g_loss_fake_id + self.lambda_rec * g_loss_rec_id + self.lambda_cls * g_loss_cls_id
                               + self.lambda_id * g_loss_id)
                loss = g_loss_fake + self.lambda_rec * g_loss_rec + self.lambda_cls * g_loss_cls + g_loss_same
                if self.use_feats:
                    loss = loss + g_loss_feat
                    loss_dict["G/loss_id_feat"] = g_loss_id_feat
                    loss_dict["G/loss_rec_feat"] = g_loss_rec_feat
                    loss_dict["G/loss_rec_id_feat"] = g_loss_rec_id_feat
                loss_dict["G/loss_fake"] = g_loss_fake
                loss_dict["G/loss_rec"] = g_loss_rec
                loss_dict["G/loss_cls"] = g_loss_cls
                loss_dict["G/loss_fake_id"] = g_loss_fake_id
                loss_dict["G/loss_rec_id"] = g_loss_rec_id
                loss_dict["G/loss_cls_id"] = g_loss_cls_id
                loss_dict["G/loss_id"] = g_loss_id
        self.loss_dict.update({k: v.detach() for k, v in loss_dict.items()})
        return loss

    # ---------------------------------------------------------------- training loop body -
    def optimizers(self):
        if self._opts is None:
            od, og = self.configure_optimizers(arenas=self.device.type == "cuda")
            self._opts = (od["optimizer"], og["optimizer"])
            self._scheds = (od["lr_scheduler"]["scheduler"], og["lr_scheduler"]["scheduler"])
        return self._opts

    def _grad_passes(self, batch, optimizer_idx: int) -> int:
        """How many backward passes of this optimizer step produce gradients of the trained network
        (the last one completes them): D step -- D's backward for (x_real, x_fake) [one paired call or
        two] plus the gradient penalty's backward-of-backward; G step -- one pass through G per
        generator call [two paired calls or four]."""
        x_real = batch[0]
        if optimizer_idx == 0:
            return 2 if (self.pair_d_calls and self._pair_fits(x_real, self.d_conv_dim // 4)) else 3
        return 2 if (self.pair_g_calls and self._pair_fits(x_real, self.g_conv_dim)) else 4

    def _reducer(self, optimizer_idx: int, arena):
        key = (optimizer_idx, id(arena), id(self.grad_sync), self.bucket_mb[optimizer_idx])
        r = self._reducers.get(optimizer_idx)
        if r is None or r[0] != key:
            r = (key, self.grad_sync.bucketed(arena, self.bucket_mb[optimizer_idx]))
            self._reducers[optimizer_idx] = r
        return r[1]

    def _flush_pending(self, *_):
        """Finish a deferred D update: wait for its gradient all-reduce, Adam step, clear the arena."""
        if self._pending is None:
            return
        handle, opt, step = self._pending
        self._pending = None
        self.grad_sync.finish(handle, "D")
        if step:
            opt.step()
        opt.arena.zero_grad()

    def _set_requires_grad(self, module, flag: bool) -> None:
        """``module.requires_grad_(flag)`` over a cached parameter list: ``nn.Module.requires_grad_`` walks the module
        tree (named_modules / named_parameters) on every call, four times per step -- about a millisecond of host time.
        The list is rebuilt when the module's parameter count changes."""
        hit = self._param_lists.get(id(module))
        if hit is None or hit[0] is not module:
            hit = (module, list(module.parameters()))
            self._param_lists[id(module)] = hit
        for p in hit[1]:
            if p.requires_grad != flag:
                p.requires_grad_(flag)

    def _run_optimizer(self, batch, batch_idx, optimizer_idx, step: bool):
        """What Lightning does per optimizer: toggle, closure (training_step + backward), step, zero_grad."""
        opt = self.optimizers()[optimizer_idx]
        if optimizer_idx == 0:
            self._flush_pending()
        mine, other = (self.D, self.G) if optimizer_idx == 0 else (self.G, self.D)
        self._set_requires_grad(other, False)
        self._set_requires_grad(mine, True)
        try:
            loss = self.training_step(batch, batch_idx, optimizer_idx)
            if loss is not None:
                arena = opt.arena if opt.arena is not None and opt.arena.intact() else None
                handle = None
                if arena is None:
                    loss.backward()
                elif self.grad_sync is not None and self.bucket_overlap and hasattr(self.grad_sync, "bucketed"):
                    # gradient spans go to the all-reduce from inside the last backward pass through `mine`
                    red = self._reducer(optimizer_idx, arena)
                    red.begin(self._grad_passes(batch, optimizer_idx))
                    mine._grad_hook = red
                    try:
                        with arena.accumulating():
                            loss.backward()
                    finally:
                        mine._grad_hook = None
                    handle = red.finish_rest()
                else:
                    with arena.accumulating():      # backward kernels add into p.grad (the arena) themselves
                        loss.backward()
                if self.grad_sync is not None:
                    if arena is not None and optimizer_idx == 0 and self.overlap_comm:
                        # D's gradient all-reduce (the large one) runs beside the generator forwards
                        # that open the G step; D is updated right before it is next used
                        if handle is None:
                            handle = self.grad_sync.allreduce_flat_async(arena.grad)
                        self._pending = (handle, opt, step)
                        return loss
                    if handle is not None:
                        self.grad_sync.finish(handle, "DG"[optimizer_idx])
                    elif arena is not None:
                        self.grad_sync.allreduce_flat(arena.grad, "DG"[optimizer_idx])
                    else:
                        self.grad_sync.allreduce([p for p in mine.parameters() if p.grad is not None])
                if step:
                    opt.step()
            if opt.arena is not None and opt.arena.intact():
                opt.arena.zero_grad()
            else:
                opt.zero_grad(set_to_none=True)
        finally:
            self._set_requires_grad(other, True)
        return loss

    def train_step(self, batch, batch_idx: int = 0, force_g: bool = False):
        """One D update and (every ``n_critic`` steps, or when ``force_g``) one G update:
        ``optimizer_step`` of ``fcd_solver.py:549-561``."""
        batch = tuple(t.to(self.device) for t in batch)
        if force_g:
            self.global_step = self.n_critic - 1 + (self.global_step // self.n_critic) * self.n_critic
        g_runs = (self.global_step + 1) % self.n_critic == 0
        x_real, c_org, c_trg = batch[0], batch[1], batch[2]
        self._in_train_step = True
        try:
            if (self.share_generator_forward and g_runs and self.G.training and self.pair_g_calls
                    and self._pair_fits(x_real, self.g_conv_dim)):
                # The reference calls G(x_real, c_trg) in the D step (fcd_solver.py:291, no graph) and again as the
                # G step's first call (:316) -- the same inputs through the same weights (G is only updated at the end
                # of the G step).  Here the call is made ONCE, as half of the G step's first paired call, with its graph
                # kept for the G step; the D step takes the detached output.  repeat_first_group gives the tracked
                # running statistics the D step's update as well, so they see five reference calls per step.
                self._set_requires_grad(self.G, True)
                with self.G.sample_groups(2), self.G.repeat_first_group(1):
                    self._shared_fake = self.G(torch.cat([x_real, x_real]), torch.cat([c_trg, c_org]))
            d_loss = self._run_optimizer(batch, batch_idx, 0, True)
            g_loss = self._run_optimizer(batch, batch_idx, 1, g_runs)
        finally:
            self._in_train_step = False
            self._shared_fake = None
        self._flush_pending()          # no G update this step: nothing used D in between
        self.global_step += 1
        if self.precision in ("fp32x2", "fp16") and (self.global_step == 1 or self.global_step % self.overflow_check_every == 0):
            # fp32x2's overflow guard (include/fcd_hip.h: fcd_set_overflow_flag): polled after the first step, where a
            # mis-wired maximum shows, and then every `overflow_check_every` steps (a poll is a host-device sync)
            ops.check_overflow(self.device)
        if self.freeze_host_objects and not self._host_frozen:
            # The interpreter's cyclic collector walks every tracked object on a full (generation 2) pass: with
            # torch imported and both networks, arenas and packed operands built that is ~78-100 ms of stopped
            # host -- four training steps of idle GPU -- and it recurs every few dozen steps (measured: bench.py's
            # `gc_pauses_in_timed_region`).  After the first complete step everything long-lived exists: collect
            # once, then move the survivors to the permanent generation so later passes only look at what the
            # training loop itself allocates.  Collection stays enabled; nothing leaks.
            gc.collect()
            gc.freeze()
            self._host_frozen = True
        return d_loss, g_loss

    # ---------------------------------------------------------------- validation --------
    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        """``fcd_solver.py:497-519``: difference map, ``> self.threshold`` (fp32 compare), confusion
        counts over pixels with ``target > 0`` (class = target - 1).  The counts are accumulated
        on the device (``fcd_threshold_hist`` with the single threshold); nothing is copied back
        per batch."""
        x_real, target = batch[0], batch[-1]
        dev = self.device
        if self._val_cm is None:
            self.sync_running_stats()
            self._val_cm = torch.zeros((2, 2), dtype=torch.int64, device=dev)
            self._val_thr = torch.tensor([self.threshold], dtype=torch.float32, device=dev)
        difference = self.compute_difference_map(x_real.to(dev))
        ops.threshold_hist(difference, target.to(dev).to(torch.uint8).contiguous(), self._val_thr, self._val_cm)

    def validation_epoch_end(self, val_step_outputs=None):
        """Confusion matrix ``cm[truth][prediction]`` and the reference's printed metrics
        (``fcd_solver.py:523-544``, ``metrics.py``); resets the accumulator."""
        if self._val_cm is not None and self.grad_sync is not None:
            # torchmetrics sums its states over ranks at compute() (fcd_solver.py:523-544)
            self.grad_sync.sum_counts(self._val_cm)
        cm = np.zeros((2, 2), dtype=np.int64) if self._val_cm is None else self._val_cm.cpu().numpy()
        self._val_cm = None
        self.check_overflow()
        # the figures of the reference's "Validation Result" line (metrics.accuracy / iou_score / f1_score)
        res = {"confusion_matrix": cm, "Accuracy": accuracy(cm), "IoU": iou_score(cm), "F1": f1_score(cm)}
        if self._scheds is not None:
            # Lightning steps both ReduceLROnPlateau schedulers on the monitored 'val/F1Score'
            # (fcd_solver.py:184-199) after every validation epoch
            for sched in self._scheds:
                sched.step(res["F1"])
        return res

    def check_overflow(self) -> None:
        """fp32x2 only: raise if any kernel since the last poll saw a scaled operand beyond 2^15 (stale max |.|)."""
        if self.precision in ("fp32x2", "fp16") and self.device.type == "cuda":
            ops.check_overflow(self.device)

    def sync_running_stats(self) -> None:
        """Data parallel: give every rank rank 0's tracked InstanceNorm running statistics (what DDP's
        ``broadcast_buffers`` keeps true in the reference) before they are read: eval-mode forwards,
        validation, threshold search, checkpointing.  A no-op on one rank."""
        if self.grad_sync is not None and hasattr(self.grad_sync, "broadcast_buffers"):
            self._flush_pending()
            self.grad_sync.broadcast_buffers([self.G])

    def state_dict(self, *args, **kwargs):
        """``G.*`` / ``D.*`` entries as in the reference's checkpoints, with the up-to-date D (a deferred
        update is finished first).  Data parallel: call :meth:`sync_running_stats` on ALL ranks before
        rank 0 saves (it is a collective, so it is not hidden in here)."""
        self._flush_pending()
        self.check_overflow()
        return super().state_dict(*args, **kwargs)

    # ---------------------------------------------------------------- inference ---------
    @torch.no_grad()
    def compute_difference_map(self, inputs):
        """mean_c(|G(x, 0) - x| / 2): translate to "no clouds" and compare (``fcd_solver.py:802-807``)."""
        c_trg = torch.zeros(inputs.shape[0], self.c_dim, device=inputs.device)
        x_fake = self.G(inputs, c_trg)
        diff, _ = ops.diffmap_threshold(x_fake, inputs.contiguous().float(), None)
        return diff

    @torch.no_grad()
    def difference_and_mask(self, inputs, threshold: float):
        """Fused difference map + ``> threshold`` (fp32 compare, uint8), ``fcd_solver.py:761-762``."""
        c_trg = torch.zeros(inputs.shape[0], self.c_dim, device=inputs.device)
        x_fake = self.G(inputs, c_trg)
        return ops.diffmap_threshold(x_fake, inputs.contiguous().float(), float(threshold))

    @torch.no_grad()
    def make_psuedo_masks(self, threshold=None, dataset: Optional[Iterable] = None, save=False, h5=True):
        """Pseudo masks for an iterable of samples ``{'image': (B,C,H,W)[, 'mask': (B,H,W)]}``.

        Same semantics as the reference (``fcd_solver.py:729-799``): ``G.eval()``, difference map,
        ``> threshold`` -> uint8, ``+1``, invalid (``mask == 0``) -> 0.  Returns the stacked uint8
        masks -- the (N, H, W) array the reference writes to ``pseudo_masks.h5:/train/masks``
        (``:785-788``); the confusion matrix it accumulates alongside (``:764-769``) is left in
        ``self.pseudo_mask_cm``.  Writing HDF5/tiff files is the caller's business (storage is out of
        the hot path)."""
        if dataset is None:
            raise ValueError("make_psuedo_masks needs an iterable of samples")
        if threshold is None:
            threshold = self.find_best_threshold(dataset=dataset)[0]
        self.G.eval()
        self.sync_running_stats()
        out, cm = [], None
        for sample in dataset:
            inputs = sample["image"].to(self.device)
            diff = self.compute_difference_map(inputs)
            target = None
            if "mask" in sample:
                target = sample["mask"].to(self.device).to(torch.uint8).contiguous()
                if cm is None:
                    cm = torch.zeros((2, 2), dtype=torch.int64, device=self.device)
            # threshold, confusion counts, +1 and invalid -> 0 in one launch; masks stay on the device
            # until the loop is done (no per-batch device-to-host copy)
            out.append(ops.pseudo_mask(diff, target, float(threshold), cm=cm if target is not None else None))
        self.pseudo_mask_cm = None if cm is None else cm.cpu().numpy()     # cm[truth][prediction], fcd_solver.py:764-769
        masks = torch.cat(out).cpu().numpy()
        self.check_overflow()
        return masks

    generate_masks = make_psuedo_masks

    @torch.no_grad()
    def find_best_threshold(self, n_thresholds=30, dataset: Optional[Iterable] = None):
        """Threshold sweep maximising macro F1 on samples ``{'image', 'mask'}`` with ground-truth
        masks (0 invalid, 1 clear, 2 cloudy): ``fcd_solver.py:680-725`` with ``metrics.py``'s
        confusion matrix / F1, first local maximum like the reference's early ``break``.
        Difference maps stay on the device; min/max and the per-threshold confusion counts are two
        device passes (``fcd_valid_minmax``, ``fcd_threshold_hist``).  The comparison
        ``preds > threshold`` is done in fp32 with the threshold rounded to fp32, as numpy 1.x
        (the reference's era) does for a float32 array against a float64 scalar."""
        self.G.eval()
        self.sync_running_stats()
        dev = self.device
        diffs, targets = [], []
        minmax = torch.tensor([float("inf"), 0.0], dtype=torch.float32, device=dev)
        for sample in dataset:
            d = self.compute_difference_map(sample["image"].to(dev))
            t = sample["mask"].to(dev).to(torch.uint8).contiguous()
            ops.valid_minmax(d, t, minmax)
            diffs.append(d)
            targets.append(t)
        lo, hi = (float(v) for v in minmax.cpu())
        if not diffs or lo == float("inf"):
            raise ValueError("find_best_threshold: no valid pixels")
        thresholds = np.linspace(start=np.float32(lo), stop=np.float32(hi), num=n_thresholds)
        thr32 = torch.from_numpy(thresholds.astype(np.float32)).to(dev)
        hist = torch.zeros((2, n_thresholds + 1), dtype=torch.int64, device=dev)
        for d, t in zip(diffs, targets):
            ops.threshold_hist(d, t, thr32, hist)
        best = best_threshold_from_hist(hist.cpu().numpy(), thresholds)
        return best, dataset

    def binarize(self, difference, threshold=0.2):
        """numpy ``>`` in the array's own dtype (``fcd_solver.py:566-567``)."""
        return (difference > threshold).astype(np.uint8)
