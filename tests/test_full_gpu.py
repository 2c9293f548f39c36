"""Full-size (BASELINE.json configs[1]: B=16, 128x128x10, 6 res blocks, D x6) parity and
size-independent properties.  Inputs and weights are regenerated from the numpy-seeded recipe
(bit-reproducible), expected values come from tests/golden/full_scalars.json.

The reference's own fp32 step is ill-conditioned at this size (its fp32 gradients differ from its
fp64 gradients by up to 3e-3 per tensor), so gradients are judged against
the fp64 oracle: the HIP error may not exceed max(1e-4, 1.25 x the reference-fp32 error) in the split-operand
modes fp32x2 (the benchmarked default) and fp32x3 (2.0 x for the exact-fp32 MFMA chains), norm-wise on two independent samples per tensor, plus an
element-wise bar (FACTORS below); the measured errors are written to
gpurun_out/full_parity_{d,g}_{precision}.json (copies under profiles/).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import relerr
from oracle import fcd_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def full(dev):
    from fcd_amd import FCDSolver
    sc = json.load(open(os.path.join(GOLD, "full_scalars.json")))
    cfg = sc["cfg"]
    gs = O.synth_generator_state(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
    ds = O.synth_discriminator_state(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"])
    batch, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])

    def make(precision="fp32"):
        s = FCDSolver(None, precision=precision)
        s.G.load_state_dict(gs)
        s.D.load_state_dict(ds)
        return s.to(dev)
    return sc, make, tuple(t.to(dev) for t in batch), alpha


BIG = np.load(os.path.join(GOLD, "full_grad_samples.npz"))
# The HIP error against the fp64 oracle may not exceed max(1e-4, factor x the reference's own fp32 error) --
# per tensor, norm-wise on two independent 4096-element samples and element-wise (max |error| / ||g64||_inf
# against the reference's figure over the whole tensor).  Factors (norm-wise, element-wise) per precision:
#   fp32x3 (the default / benchmarked mode): 1.5 and 3.0.  Both fp32 evaluations are draws from the same rounding
#     noise, amplified by the step's kinks.  Norm-wise no tensor has been further from fp64 than 1.2 x the reference's
#     own fp32 step with any set of kernels (0.89 in round 1; 1.18 on one D tensor after the summation order of the
#     generator's first convolution changed; 0.77 now): a factor of exactly 1 is a coin flip per tensor, 1.5 holds.
#     The element-wise maximum is an extreme-value statistic (one flipped ReLU mask moves a few elements by O(1) of
#     their value): worst measured 2.0 x the reference's maximum (G main.9.main.0.weight), bar 3.0;
#   fp32 (exact v_mfma_f32_32x32x2_f32 chains): 2.0 and 12 -- a sequential fp32 fma chain over K is a little less
#     accurate than torch's blocked CPU sums (worst norm-wise ratio 1.73), and the element-wise maximum is an
#     extreme-value statistic of a step with ReLU / LeakyReLU / L1 kinks: a single activation-mask flip moves a
#     few elements by O(1) of their value in either fp32 evaluation (worst ratio 9.8, D features.1.0.weight).
# The measured errors of every tensor are written to gpurun_out/full_parity_*.json.
# Round 3: the split-operand modes' bars tightened to what they need (worst measured ratios: 0.83 fp32x2 G step, 1.18
# fp32x3 D step norm-wise; 2.0 element-wise): 1.25 and 2.5 -- and the element-wise one set back to 3.0 at the end of the
# round, on this evidence: one ReLU mask element of residual block main.9 sits on its kink in fp64, and the G step's
# worst element-wise ratio is 1.8-2.0 when an evaluation leaves it on fp64's side and 2.80-2.81 (main.9.main.1.bias, both
# modes) when it does not.  Which of the two happens changes with the order of fp32 sums in a kernel five layers earlier:
# three forms of the generator's 4 x 4 stride-2 convolutions, each within 1e-6 of float64 in tests/test_conv_gpu.py (per
# tap; space to depth with the parities outermost; space to depth with the channel blocks outermost, the one kept), gave
# 1.81 / 2.80 / 2.00 under fp32x2 and 0.89 / <= 1.63 / 2.81 under fp32x3 (tools/parity_lottery.sh).  Norm-wise nothing
# moved: G step <= 0.92, D step 0.004 ... 1.18 of the reference's own fp32 error, as before.
FACTORS = {"fp32x3": (1.25, 3.0), "fp32x2": (1.25, 3.0), "fp32": (2.0, 12.0)}
OUT_DIR = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(__file__))), "gpurun_out")


def sample_errors(grad, tag, name, sc, which=""):
    """Errors vs the fp64 oracle on one of the two stored samples: (HIP norm-wise, reference-fp32 norm-wise,
    HIP max-abs / ||g64||_inf, reference-fp32 max-abs / ||g64||_inf over the whole tensor)."""
    idx = torch.from_numpy(BIG[f"{tag}.idx{which}.{name}"])
    sfx = "_2" if which == "2" else ""
    o64 = torch.from_numpy(BIG[f"{tag}.ora64{sfx}.{name}"]).double()
    r32 = torch.from_numpy(BIG[f"{tag}.ref32{sfx}.{name}"]).double()
    got = grad.detach().flatten().cpu()[idx].double()
    den = float(o64.norm()) + 1e-30
    amax = sc["grad_absmax64"][tag][name] + 1e-300
    return (float((got - o64).norm()) / den, float((r32 - o64).norm()) / den,
            float((got - o64).abs().max()) / amax, sc["ref_fp32_vs_fp64_grad_maxabs_rel"][tag][name])


def check_grads(net, tag, sc, prec, label=""):
    FACTOR, FACTOR_MAX = FACTORS[prec]
    report, worst, failures = {}, 0.0, []
    for name, p in net.named_parameters():
        entry = {}
        for which in ("", "2"):
            e, eref, emax, eref_max = sample_errors(p.grad, tag, name, sc, which)
            bar, bar_max = max(1e-4, FACTOR * eref), max(1e-4, FACTOR_MAX * eref_max)
            entry["sample" + (which or "1")] = {"hip": e, "ref_fp32": eref, "hip_maxabs": emax, "ref_fp32_maxabs": eref_max}
            worst = max(worst, e / bar, emax / bar_max)
            if e > bar:
                failures.append((name, which, "norm-wise", e, eref))
            if emax > bar_max:
                failures.append((name, which, "element-wise", emax, eref_max))
        n64 = sc[f"{tag}_grad_norm64"][name]
        nbar = max(1e-4, FACTOR * sc["ref_fp32_vs_fp64_grad_relerr"][tag][name])
        if abs(float(p.grad.double().norm()) - n64) > nbar * n64:
            failures.append((name, "", "norm", float(p.grad.double().norm()), n64))
        report[name] = entry
    try:
        os.makedirs(OUT_DIR, exist_ok=True)
        with open(os.path.join(OUT_DIR, f"full_parity_{tag}_{prec}{label}.json"), "w") as f:
            json.dump({"precision": prec, "step": tag, "factor_norm": FACTOR, "factor_elementwise": FACTOR_MAX,
                       "worst_error_over_bar": worst,
                       "errors_vs_fp64_oracle": report}, f, indent=1)
    except OSError:
        pass
    # every tensor is measured and written out first; any miss still fails the test (no survey switch)
    assert not failures, failures
    return worst


# fp32x3: weight gradients on the bf16 pipe with 3-way split operands (same bars as exact fp32)
@pytest.mark.parametrize("prec", ["fp32", "fp32x3", "fp32x2"])
def test_full_d_step(full, prec):
    sc, make, b, alpha = full
    s = make(prec)
    s.alpha_override = alpha
    s.G.requires_grad_(False)
    loss = s.training_step(b, 0, 0)
    loss.backward()
    for k, v in sc["d_losses64"].items():
        assert abs(float(s.loss_dict[k]) - v) <= 1e-4 * max(1.0, abs(v)), k
    print("full D-step", prec, ": worst error / bar =", check_grads(s.D, "d", sc, prec))


@pytest.mark.parametrize("prec", ["fp32", "fp32x3", "fp32x2"])
def test_full_g_step(full, prec):
    sc, make, b, alpha = full
    s = make(prec)
    s.D.requires_grad_(False)
    s.global_step = s.n_critic - 1
    loss = s.training_step(b, 0, 1)
    loss.backward()
    for k, v in sc["g_losses64"].items():
        assert abs(float(s.loss_dict[k]) - v) <= 1e-4 * max(1.0, abs(v)), k
    print("full G-step", prec, ": worst error / bar =", check_grads(s.G, "g", sc, prec))


@pytest.mark.parametrize("prec", ["fp32x2", "fp32x3"])
def test_full_d_step_arena_merged_weight_gradients(full, prec, monkeypatch):
    """The training path of the D step: gradients accumulated by the kernels straight into the arena, the paired batch's
    and the gradient penalty's weight gradients of every layer in ONE launch over two operand pairs (fcd_conv_wgrad2,
    Discriminator.merge_wgrad).  Same parity bars against the fp64 oracle as the autograd path; against the two-launch
    form (merge_wgrad off) only the order of fp32 sums differs."""
    from fcd_amd import ops
    sc, make, b, alpha = full
    grads, calls = {}, {}
    for merged in (True, False):
        s = make(prec)
        s.optimizers()
        s.alpha_override = alpha
        s.D.merge_wgrad = merged
        s.G.requires_grad_(False)
        n2 = [0]
        orig = ops.conv_wgrad2
        monkeypatch.setattr(ops, "conv_wgrad2", lambda *a, **k: (n2.__setitem__(0, n2[0] + 1), orig(*a, **k))[1])
        loss = s.training_step(b, 0, 0)
        with s.arenas[0].accumulating():
            loss.backward()
        monkeypatch.setattr(ops, "conv_wgrad2", orig)
        calls[merged] = n2[0]
        assert s.D._pairing.empty()                      # nothing left behind
        if merged:
            print("full D-step, arena, merged weight gradients", prec, ": worst error / bar =", check_grads(s.D, "d", sc, prec, "_arena_merged"))
        grads[merged] = {n: p.grad.clone() for n, p in s.D.named_parameters()}
    assert calls[True] == 6 and calls[False] == 0        # every layer of D took the two-pair launch
    for n in grads[True]:
        assert relerr(grads[True][n], grads[False][n]) <= 5e-6, n


@pytest.mark.parametrize("prec", ["fp32x2", "fp32x3"])
def test_full_g_step_arena_merged_weight_gradients(full, prec, monkeypatch):
    """The training path of the G step: the weight gradients of the two passes through the generator (translation and
    reconstruction) of every 3 x 3 and 4 x 4 layer in ONE launch over two operand pairs (fcd_conv_wgrad2 on the halo
    kernel, Generator.merge_wgrad), accumulated straight into the arena.  Same parity bars against the fp64 oracle as
    the autograd path; against the two-launch form only the order of fp32 sums (and, under fp32x2, the common scale of
    the two pairs) differs."""
    from fcd_amd import ops
    sc, make, b, alpha = full
    grads, calls = {}, {}
    for merged in (True, False):
        s = make(prec)
        s.optimizers()
        s.G.merge_wgrad = merged
        s.D.requires_grad_(False)
        s.global_step = s.n_critic - 1
        n2 = [0]
        orig = ops.conv_wgrad2
        monkeypatch.setattr(ops, "conv_wgrad2", lambda *a, **k: (n2.__setitem__(0, n2[0] + 1), orig(*a, **k))[1])
        loss = s.training_step(b, 0, 1)
        with s.arenas[1].accumulating():
            loss.backward()
        monkeypatch.setattr(ops, "conv_wgrad2", orig)
        calls[merged] = n2[0]
        assert s.G._pairing.empty()                      # nothing left behind
        if merged:
            print("full G-step, arena, merged weight gradients", prec, ": worst error / bar =", check_grads(s.G, "g", sc, prec, "_arena_merged"))
        grads[merged] = {n: p.grad.clone() for n, p in s.G.named_parameters()}
    assert calls[True] == 16 and calls[False] == 0       # all but the two 7 x 7 layers
    for n in grads[True]:
        assert relerr(grads[True][n], grads[False][n]) <= 2e-5, n


def test_full_steps_with_scaled_fp16_discriminator(full):
    """FCDSolver.d_scaled_fp16 (off by default: measured level with BF16X3): the discriminator on the two-piece fp16
    kernels under fp32x2, its activations' and gradients' maxima handed from convolution to convolution through
    FcdConvDesc.out_amax (epilogue or split-K reduction), through forward, backward and the gradient penalty's double
    backward.  Same bars as the default fp32x2 step; the overflow word must stay clear."""
    from fcd_amd import _lib as L
    sc, make, b, alpha = full
    s = make("fp32x2")
    s.d_scaled_fp16 = True
    s.set_precision("fp32x2")
    assert s.D.precision == L.PREC_F16X2
    s.alpha_override = alpha
    s.G.requires_grad_(False)
    s.training_step(b, 0, 0).backward()
    for k, v in sc["d_losses64"].items():
        assert abs(float(s.loss_dict[k]) - v) <= 1e-4 * max(1.0, abs(v)), k
    print("full D-step, D on the fp16 x 2 kernels: worst error / bar =", check_grads(s.D, "d", sc, "fp32x2", "_d_scaled_fp16"))
    s.check_overflow()
    s.D.zero_grad(set_to_none=True)
    s.G.requires_grad_(True)
    s.D.requires_grad_(False)
    s.global_step = s.n_critic - 1
    s.training_step(b, 0, 1).backward()
    for k, v in sc["g_losses64"].items():
        assert abs(float(s.loss_dict[k]) - v) <= 1e-4 * max(1.0, abs(v)), k
    print("full G-step, D on the fp16 x 2 kernels: worst error / bar =", check_grads(s.G, "g", sc, "fp32x2", "_d_scaled_fp16"))
    s.check_overflow()


def test_full_fp16_variant(full):
    """The reduced-precision variant of BASELINE.json configs[1] ("16-bit MFMA"; solver precision 'fp16': scaled fp16
    operands, ONE MFMA per product group in the generator, bf16 operands in the discriminator) at full size: every loss
    of the D and G steps within north_star's 1e-2 of the fp64 oracle.  Parameter gradients of this ill-conditioned
    step (the reference's own fp32 gradients are 3e-3 from its fp64 ones) are held to the
    envelope of torch's own 16-bit autocast of the same step (profiles/r2_bf16_autocast_cpu_*.json: median 5e-2 for D,
    3e-1 for G): per-tensor rel-err <= 1.0, cosine > 0.7; the measured errors are written next to the fp32 ones."""
    sc, make, b, alpha = full
    report = {}
    for tag in ("d", "g"):
        s = make("fp16")
        if tag == "d":
            s.alpha_override = alpha
            s.G.requires_grad_(False)
            s.training_step(b, 0, 0).backward()
            net, ref_losses = s.D, sc["d_losses64"]
        else:
            s.D.requires_grad_(False)
            s.global_step = s.n_critic - 1
            s.training_step(b, 0, 1).backward()
            net, ref_losses = s.G, sc["g_losses64"]
        s.check_overflow()
        for k, v in ref_losses.items():
            assert abs(float(s.loss_dict[k]) - v) <= 1e-2 * max(1.0, abs(v)), (k, float(s.loss_dict[k]), v)
        errs = {}
        for name, p in net.named_parameters():
            idx = torch.from_numpy(BIG[f"{tag}.idx.{name}"])
            o64 = torch.from_numpy(BIG[f"{tag}.ora64.{name}"]).double()
            got = p.grad.detach().flatten().cpu()[idx].double()
            e = float((got - o64).norm() / (o64.norm() + 1e-300))
            cos = float((got * o64).sum() / (got.norm() * o64.norm() + 1e-300))
            errs[name] = {"relerr": e, "cosine": cos}
            assert e <= 1.0 and cos > 0.7, (name, e, cos)
        med = sorted(v["relerr"] for v in errs.values())[len(errs) // 2]
        print("fp16 variant,", tag, "step: median gradient rel-err vs fp64", med)
        report[tag] = errs
        del s
    try:
        os.makedirs(OUT_DIR, exist_ok=True)
        json.dump(report, open(os.path.join(OUT_DIR, "full_parity_fp16.json"), "w"), indent=1)
    except OSError:
        pass


def test_full_bf16_losses(full):
    """bf16-MFMA variant (BASELINE.json configs[1], north_star tolerance 1e-2) at full size: every loss of the
    D and G steps within 1e-2 of the fp64 oracle.  Parameter gradients of this step cannot meet 1e-2 with bf16
    operands in any implementation -- torch's own CPU bf16 autocast of the same step is off by 5e-2 (D) / 3e-1
    (G) per tensor in the median (tools/bf16_autocast_cpu.py -> profiles/r2_bf16_autocast_cpu_*.json) -- so they
    are held to that envelope instead: per-tensor rel-err <= 1.0 and pointing the same way (cosine > 0.7), and
    the measured errors are written next to the fp32 ones."""
    sc, make, b, alpha = full
    report = {}
    for tag in ("d", "g"):
        s = make("bf16")
        if tag == "d":
            s.alpha_override = alpha
            s.G.requires_grad_(False)
            s.training_step(b, 0, 0).backward()
            net, ref_losses = s.D, sc["d_losses64"]
        else:
            s.D.requires_grad_(False)
            s.global_step = s.n_critic - 1
            s.training_step(b, 0, 1).backward()
            net, ref_losses = s.G, sc["g_losses64"]
        for k, v in ref_losses.items():
            assert abs(float(s.loss_dict[k]) - v) <= 1e-2 * max(1.0, abs(v)), (k, float(s.loss_dict[k]), v)
        errs = {}
        for name, p in net.named_parameters():
            idx = torch.from_numpy(BIG[f"{tag}.idx.{name}"])
            o64 = torch.from_numpy(BIG[f"{tag}.ora64.{name}"]).double()
            got = p.grad.detach().flatten().cpu()[idx].double()
            e = float((got - o64).norm() / (o64.norm() + 1e-300))
            cos = float((got * o64).sum() / (got.norm() * o64.norm() + 1e-300))
            errs[name] = {"relerr": e, "cosine": cos}
            assert e <= 1.0 and cos > 0.7, (name, e, cos)
        report[tag] = errs
        del s
    try:
        os.makedirs(OUT_DIR, exist_ok=True)
        json.dump(report, open(os.path.join(OUT_DIR, "full_parity_bf16.json"), "w"), indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("prec", ["fp32", "fp32x2"])
def test_full_masks_and_eval_forward(full, prec):
    sc, make, b, alpha = full
    s = make(prec)
    s.G.eval()
    with torch.no_grad():
        out = s.G(b[0], torch.zeros_like(b[2]))
    assert abs(float(out.double().norm()) - sc["G_eval_out_norm"]) <= 1e-4 * sc["G_eval_out_norm"]
    diff, mask = s.difference_and_mask(b[0], sc["thr"])
    assert abs(float(diff.double().sum()) - sc["diff_sum"]) <= 1e-4 * sc["diff_sum"]
    assert torch.equal(mask.cpu(), (diff.cpu() > sc["thr"]).to(torch.uint8))
    # the popcount can differ from the reference's only on pixels within rounding of the threshold
    near = int(((diff.cpu() - sc["thr"]).abs() < 1e-5).sum())
    assert abs(int(mask.sum()) - sc["mask_popcount"]) <= near + sc["diff_margin_count"]


@pytest.mark.parametrize("prec", ["fp32", "fp32x2"])
def test_full_properties(full, prec):
    """Size-independent properties: per-sample independence (what data parallelism relies on),
    determinism, invalid-pixel passthrough, and D's gradient linearity in grad_outputs."""
    sc, make, b, alpha = full
    s = make(prec)
    s.G.train()
    with torch.no_grad():
        o16 = s.G(b[0], b[2])
        o8 = s.G(b[0][:8].contiguous(), b[2][:8].contiguous())
        again = s.G(b[0], b[2])
    # InstanceNorm couples nothing across samples; only the reduction grouping depends on B
    assert float((o16[:8] - o8).abs().max()) <= 1e-5
    assert torch.equal(o16, again)             # fixed-order reductions: bitwise reproducible
    inv = b[0] == -1
    assert bool(inv.any()) and bool((o16[inv] == -1).all())
    x = b[0].clone().requires_grad_(True)
    src, _, _ = s.D(x)
    g1 = torch.autograd.grad(src, x, torch.ones_like(src), retain_graph=True)[0]
    g3 = torch.autograd.grad(src, x, 3 * torch.ones_like(src))[0]
    assert float((g3 - 3 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())


@pytest.mark.parametrize("prec", ["fp32", "fp32x2"])
def test_full_train_steps_run_and_move_weights(full, prec):
    sc, make, b, alpha = full
    s = make(prec)
    w0 = s.G.main[0].weight.detach().clone()
    d0 = s.D.conv1.weight.detach().clone()
    for _ in range(2):
        d_loss, g_loss = s.train_step(b, force_g=True)
    assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
    # Adam's first steps move every weight by about lr
    assert 1e-5 < float((s.G.main[0].weight - w0).abs().max()) < 1e-3
    assert 1e-5 < float((s.D.conv1.weight - d0).abs().max()) < 1e-3


@pytest.mark.parametrize("prec", ["fp32x2", "fp32x3"])
def test_training_is_bitwise_reproducible(full, prec):
    """Ten full-size training steps, twice, from the same state: identical bits in every weight and
    running statistic.  All reductions run in a fixed order and nothing in the training path uses
    floating-point atomics, so any difference would mean a race (LDS-DMA waits, stream ordering of
    the arena updates) -- tools/soak_probe.py is the 400-step version of this."""
    import hashlib
    sc, make, b, alpha = full

    def run():
        s = make(prec)
        s.alpha_override = alpha
        for _ in range(10):
            d_loss, g_loss = s.train_step(b, force_g=True)
        assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
        h = hashlib.sha256()
        for net in (s.G, s.D):
            for v in net.state_dict().values():
                h.update(v.detach().cpu().numpy().tobytes())
        return h.hexdigest()

    assert run() == run()


@pytest.mark.parametrize("act", ["silu", "mish"])
def test_full_width_smooth_networks_fp32x2(dev, act):
    """Full-width networks with a smooth activation in both (fixed_point_gan.py:13-16) under fp32x2: the activation
    codes travel through the fused max |.| producers (InstanceNorm apply with amax; the smooth derivatives keep the
    three-kernel InstanceNorm backward); the G-step's parameter gradients of fp32x2 against fp32x3 and of both against
    the exact-fp32 mode of the same library."""
    from fcd_amd import FCDSolver
    from fcd_amd.synth import synth_batch
    batch, _ = synth_batch(8, 128, 10, seed=17)
    b = tuple(t.to(dev) for t in batch)
    grads = {}
    for mode in ("fp32", "fp32x3", "fp32x2"):
        torch.manual_seed(11)
        s = FCDSolver(None, act_G=act, act_D=act, batch_size=8, precision=mode).to(dev)
        s.D.requires_grad_(False)
        s.global_step = s.n_critic - 1
        loss = s.training_step(b, 0, 1)
        loss.backward()
        assert torch.isfinite(loss)
        grads[mode] = {k: p.grad.detach().double().cpu() for k, p in s.G.named_parameters()}
    def worst(a, b_):
        return max(float((grads[a][k] - g).norm() / (g.norm() + 1e-30)) for k, g in grads[b_].items())

    w = {"x2_vs_x3": worst("fp32x2", "fp32x3"), "x2_vs_fp32": worst("fp32x2", "fp32"), "x3_vs_fp32": worst("fp32x3", "fp32")}
    print("smooth", act, "worst rel. difference of G-step gradients between modes:", w)
    # random-init networks are ill-conditioned and the L1 terms of the G loss keep their sign kinks:
    # measured 1e-4 .. 1.2e-3 between any two of the three modes, with a different mode the outlier for SiLU and for
    # Mish -- what this test pins is that the smooth-activation plumbing of fp32x2 runs at full width and lands in the
    # same band
    assert max(w.values()) <= 5e-3, w
