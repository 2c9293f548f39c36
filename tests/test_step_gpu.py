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


@pytest.mark.parametrize("prec", ["fp32", "fp32x3", "fp32x2"])
def test_d_step_losses_and_grads(dev, tiny, prec):
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    s.set_precision(prec)
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


@pytest.mark.parametrize("prec", ["fp32", "fp32x3", "fp32x2"])
def test_g_step_losses_and_grads(dev, tiny, prec):
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    s.set_precision(prec)
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


def _oracle_two_steps(z, sc, lr):
    from oracle import fcd_oracle as O
    gs = {k[2:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("G.")}
    ds = {k[2:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("D.")}
    t = lambda k: torch.from_numpy(z[k])
    batch = (t("x"), t("c_org"), t("c_trg"), t("label_org"), t("label_trg"))
    alpha = t("alpha")
    st = {}
    losses = {}
    for step in (1, 2):
        ns = {}
        _, ld, dg = O.d_step(gs, ds, batch, alpha, new_stats=ns)
        gs.update(ns)
        for k, g in dg.items():
            m, v = st.get(("d", k), (torch.zeros_like(g), torch.zeros_like(g)))
            ds[k], m, v = O.adam_update(ds[k], g, m, v, step, lr=lr)
            st[("d", k)] = (m, v)
        ns = {}
        _, lg, gg = O.g_step(gs, ds, batch, new_stats=ns)
        gs.update(ns)
        for k, g in gg.items():
            m, v = st.get(("g", k), (torch.zeros_like(g), torch.zeros_like(g)))
            gs[k], m, v = O.adam_update(gs[k], g, m, v, step, lr=lr)
            st[("g", k)] = (m, v)
        losses = {**ld, **lg}
    return gs, ds, losses


def test_two_train_steps_match_oracle(dev, tiny):
    """Two full train steps (D update, G update, both Adam steps, running statistics) against
    the CPU oracle: the second step only matches if the first one's updates really reached the
    kernels (packed weight copies refreshed, arena views intact)."""
    z, sc = tiny
    lr = 1e-3
    s = build(z, sc["cfg"], dev)
    s.d_lr = s.g_lr = lr
    s.alpha_override = torch.from_numpy(z["alpha"])
    b = batch_of(z, dev)
    for _ in range(2):
        s.train_step(b, force_g=True)
    torch.cuda.synchronize()
    assert all(a is not None and a.intact() for a in s.arenas)
    gs, ds, losses = _oracle_two_steps(z, sc, lr)
    for k, v in losses.items():
        assert abs(float(s.loss_dict[k]) - float(v)) <= 2e-4 * max(1.0, abs(float(v))), (k, float(s.loss_dict[k]), float(v))
    # Adam's first steps are ~lr*sign(g): an element whose gradient is at rounding level can land
    # on the other side, so parameters are compared with a bound of a few such elements
    worst = 0.0
    for net, sd in ((s.G, gs), (s.D, ds)):
        for name, t in net.state_dict().items():
            if not t.is_floating_point():
                continue
            worst = max(worst, relerr(t, sd[name]))
    print("two-step worst state rel-err", worst)
    assert worst <= 5e-3


@pytest.mark.parametrize("prec", ["fp32", "fp32x2"])
def test_shared_generator_forward_equals_reference_sequence(dev, tiny, prec):
    """train_step makes the generator call G(x_real, c_trg), which the reference computes twice per step with the same
    weights (fcd_solver.py:291 in the D step, :316 in the G step), once, and leaves out D(x_reconst_id) (:358), whose
    outputs nothing reads unless use_feats.  Against the reference's own call sequence (both switches off): the same
    losses, weights and running statistics after three steps, up to the rounding of one 16-sample call against half of
    a 32-sample call (running statistics: every reference call still counts -- five momentum updates per step)."""
    z, sc = tiny
    states = {}
    for share in (True, False):
        s = build(z, sc["cfg"], dev)
        s.set_precision(prec)
        s.share_generator_forward = share
        s.skip_unused_d_forward = share
        s.alpha_override = torch.from_numpy(z["alpha"])
        b = batch_of(z, dev)
        for _ in range(3):
            s.train_step(b, force_g=True)
        torch.cuda.synchronize()
        assert s._shared_fake is None
        states[share] = ({k: v.detach().clone() for k, v in s.state_dict().items()},
                         {k: float(v) for k, v in s.loss_dict.items()})
    (sd1, l1), (sd0, l0) = states[True], states[False]
    for k in l0:
        assert abs(l1[k] - l0[k]) <= 1e-4 * max(1.0, abs(l0[k])), (k, l1[k], l0[k])
    worst, worst_stats = 0.0, 0.0
    for k, v in sd0.items():
        if not v.is_floating_point():
            continue
        e = relerr(sd1[k], v)
        if "running" in k:
            worst_stats = max(worst_stats, e)
        else:
            worst = max(worst, e)
    print("shared forward vs reference sequence:", prec, "weights", worst, "running statistics", worst_stats)
    assert worst_stats <= 1e-5            # an update more or less would show at the 1e-1 level
    assert worst <= 5e-3                  # Adam's first steps are ~lr * sign(g): rounding-level gradients may flip


@pytest.mark.parametrize("prec", ["fp32", "fp32x2"])
def test_side_stream_weight_gradients_are_the_same_bits(dev, tiny, prec):
    """The weight gradients run on a second HIP stream beside the data-gradient chain (ops.SideWork); the same kernels
    accumulate into the same arena in the same order, so three steps must leave the same bits in every weight and
    running statistic as the one-stream order."""
    z, sc = tiny
    states = {}
    for side in (True, False):
        s = build(z, sc["cfg"], dev)
        s.set_precision(prec)
        s.G.wgrad_side_stream = s.D.wgrad_side_stream = side
        s.alpha_override = torch.from_numpy(z["alpha"])
        b = batch_of(z, dev)
        for _ in range(3):
            s.train_step(b, force_g=True)
        torch.cuda.synchronize()
        states[side] = {k: v.detach().clone() for k, v in s.state_dict().items()}
    for k, v in states[False].items():
        assert torch.equal(states[True][k], v), k


def test_arena_gradients_match_fixtures(dev, tiny):
    """Same D-step gradients when the backward kernels accumulate straight into the gradient
    arena (the training path) instead of going through autograd (the other tests)."""
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    s.optimizers()                       # creates the arenas
    assert s.arenas[0].intact() and s.arenas[1].intact()
    b = batch_of(z, dev)
    s.alpha_override = torch.from_numpy(z["alpha"])
    s.G.requires_grad_(False)
    loss = s.training_step(b, 0, 0)
    with s.arenas[0].accumulating():
        loss.backward()
    for name, p in s.D.named_parameters():
        assert relerr(p.grad, torch.from_numpy(z["dgrad." + name])) <= TOL, name
    s.G.requires_grad_(True)
    s.D.requires_grad_(False)
    s2 = build(z, sc["cfg"], dev)
    s2.optimizers()
    s2.D.requires_grad_(False)
    s2.global_step = s2.n_critic - 1
    loss = s2.training_step(b, 0, 1)
    with s2.arenas[1].accumulating():
        loss.backward()
    for name, p in s2.G.named_parameters():
        assert relerr(p.grad, torch.from_numpy(z["ggrad." + name])) <= TOL, name


class _FakeSync:
    """Stands in for parallel.GradSync on one GPU: records the calls, reduces nothing."""
    world = 2

    def __init__(self):
        self.async_calls = self.sync_calls = self.finished = 0

    def allreduce_flat_async(self, flat):
        self.async_calls += 1
        return [("handle", flat)]

    def finish(self, handle, tag=""):
        self.finished += len(handle)

    def allreduce_flat(self, flat, tag=""):
        self.sync_calls += 1

    def allreduce(self, params):
        self.sync_calls += 1


def test_deferred_d_update_equals_immediate(dev, tiny):
    """Data-parallel schedule: D's all-reduce is started after its backward and D is updated right
    before its next use (inside the G step).  Must give bit-identical weights to the immediate
    update, also on steps that skip the G update."""
    z, sc = tiny
    b = batch_of(z, dev)
    states = []
    for mode in ("none", "overlap", "sync"):
        s = build(z, sc["cfg"], dev)
        s.alpha_override = torch.from_numpy(z["alpha"])
        if mode != "none":
            s.grad_sync = _FakeSync()
            s.overlap_comm = mode == "overlap"
        for it in range(3):
            s.train_step(b, force_g=(it != 1))     # middle step: D update only
        torch.cuda.synchronize()
        if mode == "overlap":
            assert s.grad_sync.async_calls == 3 and s.grad_sync.finished == 3 and s._pending is None
        states.append({k: v.clone() for k, v in list(s.G.state_dict().items()) + list(s.D.state_dict().items())})
    for other in states[1:]:
        for k, v in states[0].items():
            assert torch.equal(v, other[k]), k


def test_bf16_variant_losses(dev, tiny):
    """bf16-MFMA variant (operands rounded to bf16, fp32 accumulate): every loss of the D and G
    steps within 1e-2 of the fp32 reference fixtures (BASELINE.json north_star tolerance for bf16).
    Parameter gradients are NOT held to 1e-2: the step amplifies operand rounding (ReLU / L1 kinks,
    four chained generator passes) to 3e-2..3e-1 per tensor in any bf16 implementation (DESIGN.md
    section 6); they are checked for direction only."""
    z, sc = tiny
    b = batch_of(z, dev)
    s = build(z, sc["cfg"], dev)
    s.set_precision("bf16")
    s.alpha_override = torch.from_numpy(z["alpha"])
    s.G.requires_grad_(False)
    s.training_step(b, 0, 0).backward()
    for k, v in sc["d_losses"].items():
        assert abs(float(s.loss_dict[k]) - v) <= 1e-2 * max(1.0, abs(v)), k
    for name, p in s.D.named_parameters():
        ref = torch.from_numpy(z["dgrad." + name]).to(dev)
        cos = float((p.grad * ref).sum() / (p.grad.norm() * ref.norm() + 1e-30))
        assert cos > 0.9, (name, cos)
    s = build(z, sc["cfg"], dev)
    s.set_precision("bf16")
    s.D.requires_grad_(False)
    s.global_step = s.n_critic - 1
    s.training_step(b, 0, 1).backward()
    for k, v in sc["g_losses"].items():
        assert abs(float(s.loss_dict[k]) - v) <= 1e-2 * max(1.0, abs(v)), k
    for name, p in s.G.named_parameters():
        ref = torch.from_numpy(z["ggrad." + name]).to(dev)
        cos = float((p.grad * ref).sum() / (p.grad.norm() * ref.norm() + 1e-30))
        assert cos > 0.8, (name, cos)


# --------------------------------------------------------------------------- #
# optional configurations: use_feats, interm_non_act, ReLU discriminator
# --------------------------------------------------------------------------- #
VARIANTS = {
    # name: (act_g, act_d, n_feat_layers, interm_non_act, use_feats) -- oracle/make_golden.py VARIANTS
    "feats": ("relu", "lrelu", 2, False, True),
    "non_act": ("relu", "lrelu", 2, True, True),
    "non_act_allfeat": ("relu", "lrelu", 4, True, True),
    "relu_d": ("lrelu", "relu", 2, False, False),
    "silu_g": ("silu", "lrelu", 4, False, False),
    "mish_g": ("mish", "lrelu", 4, False, True),
    "silu_d": ("relu", "silu", 2, False, True),
    "mish_d": ("silu", "mish", 2, True, True),
}


def build_variant(name, cfg, dev):
    import re
    from fcd_amd import FCDSolver
    from oracle import fcd_oracle as O
    act_g, act_d, n_feat, ina, feats = VARIANTS[name]
    s = FCDSolver(None, c_dim=cfg["c_dim"], image_size=cfg["size"], g_conv_dim=cfg["g_conv_dim"],
                  d_conv_dim=cfg["d_conv_dim"], g_repeat_num=cfg["g_repeat"], d_repeat_num=cfg["d_repeat"],
                  num_channels=cfg["bands"], batch_size=cfg["batch"], act_G=act_g, act_D=act_d, n_feat_layers=n_feat,
                  interm_non_act=ina, use_feats=feats)
    gs = O.synth_generator_state(cfg["g_conv_dim"], cfg["c_dim"], cfg["g_repeat"], cfg["bands"])
    ds = O.synth_discriminator_state(cfg["size"], cfg["d_conv_dim"], cfg["c_dim"], cfg["d_repeat"], cfg["bands"],
                                     n_feature_layers=n_feat)
    if ina:     # the reference's keys: Sequential(act, conv) puts the convolution at index 1
        ds = {re.sub(r"^(features|main)\.(\d+)\.0\.", r"\1.\2.1.", k): v for k, v in ds.items()}
    s.G.load_state_dict(gs)
    s.D.load_state_dict(ds)          # strict: the key layout must be the reference's
    return s.to(dev)


def _backward(s, loss, optimizer_idx, arena):
    if arena:
        with s.arenas[optimizer_idx].accumulating():     # kernels add into p.grad (the arena) themselves
            loss.backward()
    else:
        loss.backward()


@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("prec", ["fp32", "fp32x3", "fp32x2", "fp32-arena", "fp32x2-arena"])
def test_variant_steps(dev, tiny, name, prec):
    """prec 'fp32-arena': gradients accumulated by the kernels straight into the flat arenas
    (the training path) instead of being returned through autograd."""
    from oracle import fcd_oracle as O
    z, sc = tiny
    v = np.load(os.path.join(GOLD, "tiny_variants.npz"))
    cfg = sc["cfg"]
    batch, alpha = O.synth_batch(cfg["batch"], cfg["size"], cfg["bands"])
    b = tuple(t.to(dev) for t in batch)
    arena = prec.endswith("-arena")
    prec = prec.split("-")[0]
    s = build_variant(name, cfg, dev)
    if arena:
        s.optimizers()
    s.set_precision(prec)
    with torch.no_grad():
        src, _, feat = s.D(b[0])
    assert relerr(src, torch.from_numpy(v[f"{name}.D_src"])) <= TOL
    assert relerr(feat[0], torch.from_numpy(v[f"{name}.D_feat"])) <= TOL
    # D step
    s.alpha_override = alpha
    s.G.requires_grad_(False)
    _backward(s, s.training_step(b, 0, 0), 0, arena)
    for k in [k for k in v.files if k.startswith(f"{name}.dloss.")]:
        ref = float(v[k])
        assert abs(float(s.loss_dict[k.split(".dloss.")[1]]) - ref) <= TOL * max(1.0, abs(ref)), k
    worst = 0.0
    for pname, p in s.D.named_parameters():
        e = relerr(p.grad if p.grad is not None else torch.zeros_like(p), torch.from_numpy(v[f"{name}.dgrad.{pname}"]))
        worst = max(worst, e)
        assert e <= TOL, (name, pname, e)
    # G step (fresh modules: running statistics as in the fixture)
    s = build_variant(name, cfg, dev)
    if arena:
        s.optimizers()
    s.set_precision(prec)
    s.D.requires_grad_(False)
    s.global_step = s.n_critic - 1
    _backward(s, s.training_step(b, 0, 1), 1, arena)
    gl = [k for k in v.files if k.startswith(f"{name}.gloss.")]
    assert len(gl) == (10 if VARIANTS[name][4] else 7)
    for k in gl:
        ref = float(v[k])
        assert abs(float(s.loss_dict[k.split(".gloss.")[1]]) - ref) <= TOL * max(1.0, abs(ref)), k
    for pname, p in s.G.named_parameters():
        e = relerr(p.grad, torch.from_numpy(v[f"{name}.ggrad.{pname}"]))
        worst = max(worst, e)
        assert e <= TOL, (name, pname, e)
    print(f"variant {name} [{prec}] worst grad rel-err {worst:.2e}")


# --------------------------------------------------------------------------- #
# other shapes against the CPU oracle (float64): the constructor defaults of the reference modules
# (c_dim 5, 3 channels: fixed_point_gan.py:39,102), a batch of one, odd channel counts
# --------------------------------------------------------------------------- #
SHAPES = {
    # name: (batch, size, bands, c_dim, g_conv_dim, d_conv_dim, g_repeat, d_repeat)
    "stargan_defaults": (3, 32, 3, 5, 8, 8, 2, 3),
    "batch_of_one": (1, 32, 10, 1, 8, 16, 1, 4),
    "seven_bands": (2, 64, 7, 2, 12, 8, 1, 5),
}


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("prec", ["fp32", "fp32x3", "fp32x2"])
def test_other_shapes_vs_oracle(dev, name, prec):
    from fcd_amd import FCDSolver
    from oracle import fcd_oracle as O
    b, size, bands, c_dim, gcd, dcd, grep, drep = SHAPES[name]
    gs = O.synth_generator_state(gcd, c_dim, grep, bands)
    ds = O.synth_discriminator_state(size, dcd, c_dim, drep, bands)
    (x, _, _, _, _), alpha = O.synth_batch(b, size, bands, seed=11)
    rng = np.random.default_rng(12)
    label_org = torch.from_numpy((rng.random((b, c_dim)) < 0.5).astype(np.float32))
    label_trg = label_org[torch.from_numpy(rng.permutation(b))]
    batch = (x, label_org.clone(), label_trg.clone(), label_org, label_trg)
    # Truth = the oracle in float64.  (torch's CPU fp32 evaluation of the same graph is itself off
    # by up to 3e-3 per tensor on 'seven_bands' -- ReLU / L1 kinks, DESIGN section 2 -- while the HIP
    # path is within 3e-6 of float64: tools/shape_probe.py.)
    gs64, ds64 = O.to_dtype(gs, torch.float64), O.to_dtype(ds, torch.float64)
    b64 = tuple(t.double() for t in batch)
    _, dl, dg = O.d_step(gs64, ds64, b64, alpha.double(), new_stats={})
    ns = {}
    _, gl, gg = O.g_step(gs64, ds64, b64, new_stats=ns)

    def make():
        s = FCDSolver(None, c_dim=c_dim, image_size=size, g_conv_dim=gcd, d_conv_dim=dcd, g_repeat_num=grep,
                      d_repeat_num=drep, num_channels=bands, batch_size=b)
        s.G.load_state_dict(gs)
        s.D.load_state_dict(ds)
        s.to(dev)
        s.set_precision(prec)
        return s

    gb = tuple(t.to(dev) for t in batch)
    s = make()
    s.alpha_override = alpha
    s.G.requires_grad_(False)
    s.training_step(gb, 0, 0).backward()
    for k, v in dl.items():
        assert abs(float(s.loss_dict[k]) - float(v)) <= TOL * max(1.0, abs(float(v))), k
    worst = 0.0
    for pname, p in s.D.named_parameters():
        e = relerr(p.grad if p.grad is not None else torch.zeros_like(p), dg[pname])
        worst = max(worst, e)
        assert e <= TOL, (name, pname, e)
    s = make()
    s.D.requires_grad_(False)
    s.global_step = s.n_critic - 1
    s.training_step(gb, 0, 1).backward()
    for k, v in gl.items():
        assert abs(float(s.loss_dict[k]) - float(v)) <= TOL * max(1.0, abs(float(v))), k
    for pname, p in s.G.named_parameters():
        e = relerr(p.grad, gg[pname])
        worst = max(worst, e)
        assert e <= TOL, (name, pname, e)
    for k, v in ns.items():
        assert relerr(s.G.state_dict()[k], v) <= TOL, k
    print(f"shape {name} [{prec}] worst grad rel-err vs oracle {worst:.2e}")


def test_paired_calls_equal_separate_calls(dev, tiny):
    """The solver runs the G step's four generator calls as two paired-batch calls and the D step's
    D(x_real), D(x_fake) as one (pair_g_calls / pair_d_calls): losses, gradients and the tracked
    running statistics must be those of the reference's separate calls."""
    z, sc = tiny
    b = batch_of(z, dev)
    out = {}
    for paired in (True, False):
        s = build(z, sc["cfg"], dev)
        s.pair_g_calls = s.pair_d_calls = paired
        s.alpha_override = torch.from_numpy(z["alpha"])
        s.G.requires_grad_(False)
        s.training_step(b, 0, 0).backward()
        dl = {k: float(v) for k, v in s.loss_dict.items()}
        dg = {n: p.grad.clone() for n, p in s.D.named_parameters()}
        s.G.requires_grad_(True)
        s.D.requires_grad_(False)
        s.global_step = s.n_critic - 1
        s.training_step(b, 0, 1).backward()
        gl = {k: float(v) for k, v in s.loss_dict.items()}
        gg = {n: p.grad.clone() for n, p in s.G.named_parameters()}
        stats = {k: v.clone() for k, v in s.G.state_dict().items() if "running_" in k}
        out[paired] = (dl, dg, gl, gg, stats)
    (dl1, dg1, gl1, gg1, st1), (dl0, dg0, gl0, gg0, st0) = out[True], out[False]
    for k in gl0:
        assert abs(gl1[k] - gl0[k]) <= 2e-6 * max(1.0, abs(gl0[k])), k
    for k in dg0:
        assert relerr(dg1[k], dg0[k]) <= 2e-5, k
    for k in gg0:
        assert relerr(gg1[k], gg0[k]) <= 2e-5, k
    for k in st0:      # same updates in the same order: bit-identical
        assert torch.equal(st1[k], st0[k]), k


def test_all_nodata_sample_and_constant_channels(dev):
    """Domain edge cases: one sample of the batch is entirely nodata (every band -1: constant input,
    zero variance in the first InstanceNorm's statistics path, output forced to -1 everywhere) and
    another has a constant band.  Forward, losses and gradients against the float64 oracle."""
    from fcd_amd import FCDSolver
    from oracle import fcd_oracle as O
    b, size, bands, c_dim, gcd, dcd, grep, drep = 3, 32, 10, 1, 8, 8, 2, 3
    gs = O.synth_generator_state(gcd, c_dim, grep, bands)
    ds = O.synth_discriminator_state(size, dcd, c_dim, drep, bands)
    (x, c_org, c_trg, l_org, l_trg), alpha = O.synth_batch(b, size, bands, seed=21)
    x = x.clone()
    x[1] = -1.0                     # a patch without any valid pixel
    x[2, 4] = 0.25                  # a constant band
    batch = (x, c_org, c_trg, l_org, l_trg)
    gs64, ds64 = O.to_dtype(gs, torch.float64), O.to_dtype(ds, torch.float64)
    b64 = tuple(t.double() for t in batch)
    with torch.no_grad():
        ref_out = O.generator_forward(gs64, b64[0], b64[2], True, "relu", {})
    _, dl, dg = O.d_step(gs64, ds64, b64, alpha.double(), new_stats={})
    _, gl, gg = O.g_step(gs64, ds64, b64, new_stats={})

    def make():
        s = FCDSolver(None, c_dim=c_dim, image_size=size, g_conv_dim=gcd, d_conv_dim=dcd, g_repeat_num=grep,
                      d_repeat_num=drep, num_channels=bands, batch_size=b)
        s.G.load_state_dict(gs)
        s.D.load_state_dict(ds)
        return s.to(dev)

    gb = tuple(t.to(dev) for t in batch)
    s = make()
    with torch.no_grad():
        out = s.G(gb[0], gb[2])
    assert bool((out[1] == -1).all()) and torch.isfinite(out).all()
    assert relerr(out, ref_out) <= TOL
    s = make()
    s.alpha_override = alpha
    s.G.requires_grad_(False)
    s.training_step(gb, 0, 0).backward()
    for k, v in dl.items():
        assert abs(float(s.loss_dict[k]) - float(v)) <= TOL * max(1.0, abs(float(v))), k
    for pname, p in s.D.named_parameters():
        assert relerr(p.grad, dg[pname]) <= TOL, pname
    s = make()
    s.D.requires_grad_(False)
    s.global_step = s.n_critic - 1
    s.training_step(gb, 0, 1).backward()
    for k, v in gl.items():
        assert abs(float(s.loss_dict[k]) - float(v)) <= TOL * max(1.0, abs(float(v))), k
    for pname, p in s.G.named_parameters():
        assert torch.isfinite(p.grad).all() and relerr(p.grad, gg[pname]) <= TOL, pname


# --------------------------------------------------------------------------- #
# eval-mode folding, arena safety (round-1 advisor findings)
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("prec", ["fp32", "fp32x3", "fp32x2"])
def test_folded_eval_equals_unfolded_eval(dev, tiny, prec):
    """Under torch.no_grad() the eval-mode generator folds the tracked InstanceNorms into the
    convolutions (ConvParams.folded); with grad enabled and parameters requiring grad it takes the
    unfolded stats_from_running + in_apply path.  Same outputs and masks, and a fold must not survive
    a kernel-side change of the running statistics (a train-mode forward between two evals)."""
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    s.set_precision(prec)
    b = batch_of(z, dev)
    c0 = torch.zeros_like(b[2])

    def both():
        s.G.eval()
        with torch.no_grad():
            folded = s.G(b[0], c0)
        assert any("folded" in m._packed for m in s.G.modules() if hasattr(m, "_packed"))
        with torch.enable_grad():
            unfolded = s.G(b[0], c0).detach()
        return folded, unfolded

    f1, u1 = both()
    assert relerr(f1, u1) <= 1e-5
    assert relerr(f1, torch.from_numpy(z["G_eval_out"])) <= TOL
    d_f, m_f = ops_diff(f1, b[0], sc["thr"])
    d_u, m_u = ops_diff(u1, b[0], sc["thr"])
    far = (d_u - sc["thr"]).abs() > 1e-5
    assert bool((m_f[far] == m_u[far]).all())
    # running statistics move (weights unchanged: a D-only phase), then eval again
    s.G.train()
    with torch.no_grad():
        for _ in range(3):
            s.G(b[0] * 0.5, b[2])
    f2, u2 = both()
    assert relerr(f2, u2) <= 1e-5                      # a stale fold would reproduce f1 instead
    assert relerr(f2, f1) > 1e-4
    # and torch-side writes (load_state_dict) invalidate it as well
    s.G.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("G.")})
    f3, u3 = both()
    assert relerr(f3, f1) <= 1e-6 and relerr(f3, u3) <= 1e-5


def ops_diff(x_fake, x, thr):
    from fcd_amd import ops
    return ops.diffmap_threshold(x_fake.contiguous(), x.contiguous(), float(thr))


def test_arena_untouched_by_autograd_grad(dev, tiny):
    """A reference-style gradient_penalty (no input_grad_only) on arena-backed parameters:
    torch.autograd.grad(..., create_graph=True) must leave p.grad alone; only a backward inside
    arena.accumulating() writes the arena directly."""
    z, sc = tiny
    s = build(z, sc["cfg"], dev)
    s.optimizers()
    ar = s.arenas[0]
    assert ar.intact() and not ar.direct
    def penalty():
        x = batch_of(z, dev)[0].clone().requires_grad_(True)
        src, _, _ = s.D(x)
        dydx = torch.autograd.grad(outputs=src, inputs=x, grad_outputs=torch.ones_like(src), retain_graph=True,
                                   create_graph=True, only_inputs=True)[0]
        torch.cuda.synchronize()
        assert float(ar.grad.abs().sum()) == 0.0          # the first-order pass left the arena alone
        return torch.mean((torch.sqrt(torch.sum(dydx.view(dydx.size(0), -1) ** 2, dim=1)) - 1) ** 2)

    # outside accumulating(): gradients travel through autograd into p.grad (the arena views)
    penalty().backward()
    via_autograd = ar.grad.clone()
    assert float(via_autograd.abs().sum()) > 0 and ar.intact()
    ar.zero_grad()
    gp = penalty()
    with ar.accumulating():
        gp.backward()
    assert relerr(ar.grad, via_autograd) <= 1e-6 and not ar.direct


def test_adam_arena_state_dict_round_trip(dev, tiny):
    """FusedAdam with an arena: state saved after two steps and loaded into a fresh optimizer
    continues with the same bits (the flat moment buffers are seeded from the loaded state)."""
    z, sc = tiny
    b = batch_of(z, dev)

    def fresh():
        s = build(z, sc["cfg"], dev)
        s.alpha_override = torch.from_numpy(z["alpha"])
        return s

    a = fresh()
    for _ in range(2):
        a.train_step(b, force_g=True)
    a._flush_pending()
    import copy
    saved = copy.deepcopy({"G": a.G.state_dict(), "D": a.D.state_dict(),       # state_dict() aliases live storage
                           "opt": [o.state_dict() for o in a.optimizers()]})
    a.train_step(b, force_g=True)
    c = fresh()
    c.G.load_state_dict(saved["G"])
    c.D.load_state_dict(saved["D"])
    for o, sd in zip(c.optimizers(), saved["opt"]):
        o.load_state_dict(sd)
    c.global_step = 2
    c.train_step(b, force_g=True)
    torch.cuda.synchronize()
    for net_a, net_c in ((a.G, c.G), (a.D, c.D)):
        for (k, va), vc in zip(net_a.state_dict().items(), net_c.state_dict().values()):
            assert torch.equal(va, vc), k
