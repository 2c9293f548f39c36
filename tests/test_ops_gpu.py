"""Op-level parity of the HBM-bound HIP kernels (through the C ABI) against torch on CPU."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("shape", [(2, 8, 5, 7), (16, 64, 32, 32), (3, 256, 16, 16), (2, 64, 128, 128)])
@pytest.mark.parametrize("slope", [0.0, 1.0])
def test_instnorm_forward_backward(dev, shape, slope):
    from fcd_amd import ops
    torch.manual_seed(0)
    n, c, h, w = shape
    z = (torch.randn(shape) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(c)).requires_grad_(True)
    beta = (0.2 * torch.randn(c)).requires_grad_(True)
    res = torch.randn(shape)
    gy = torch.randn(shape)
    y0 = F.instance_norm(z, None, None, gamma, beta, True, 0.1, 1e-5)
    y_ref = (F.relu(y0) if slope == 0.0 else y0) + res
    y_ref.backward(gy)
    zd = nhwc(z.detach()).to(dev)
    mean, rstd, varu = ops.instnorm_stats(zd, want_var=True)
    assert relerr(mean, z.detach().mean((2, 3))) <= 1e-5
    assert relerr(varu, z.detach().var((2, 3), unbiased=True)) <= 1e-5
    y = ops.instnorm_apply(zd, mean, rstd, gamma.detach().to(dev), beta.detach().to(dev), slope, residual=nhwc(res).to(dev))
    assert relerr(y.cpu(), nhwc(y_ref.detach())) <= 1e-5
    # backward: the activation mask comes from the pre-residual output
    y_pre = ops.instnorm_apply(zd, mean, rstd, gamma.detach().to(dev), beta.detach().to(dev), slope)
    dz, dg, db = ops.instnorm_bwd(nhwc(gy).to(dev), y_pre, zd, mean, rstd, gamma.detach().to(dev), slope, False)
    assert relerr(dz.cpu(), nhwc(z.grad)) <= 2e-5
    assert relerr(dg, gamma.grad) <= 2e-5 and relerr(db, beta.grad) <= 2e-5
    # mask recomputed from z (beta given, no y): the same bits as with the mask taken from y
    dz2, dg2, db2 = ops.instnorm_bwd(nhwc(gy).to(dev), None, zd, mean, rstd, gamma.detach().to(dev), slope, False,
                                     beta=beta.detach().to(dev))
    assert torch.equal(dz2, dz) and torch.equal(dg2, dg) and torch.equal(db2, db)
    assert bool(((y_pre > 0) == (ops.instnorm_apply(zd, mean, rstd, gamma.detach().to(dev), beta.detach().to(dev), 1.0) > 0)).all())


def test_instnorm_running_stats_and_eval(dev):
    from fcd_amd import ops
    torch.manual_seed(1)
    n, c, h, w = 4, 16, 8, 8
    m = torch.nn.InstanceNorm2d(c, affine=True, track_running_stats=True)
    with torch.no_grad():
        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.2)
    z = torch.randn(n, c, h, w) * 1.5
    rm, rv = m.running_mean.clone().to(dev), m.running_var.clone().to(dev)
    m.train(); m(z)
    mean, rstd, varu = ops.instnorm_stats(nhwc(z).to(dev), want_var=True)
    ops.instnorm_update_running(rm, rv, mean, varu)
    assert relerr(rm, m.running_mean) <= 1e-5 and relerr(rv, m.running_var) <= 1e-5
    m.eval()
    ref = m(z)
    mean, rstd = ops.instnorm_stats_from_running(rm, rv, n)
    y = ops.instnorm_apply(nhwc(z).to(dev), mean, rstd, m.weight.detach().to(dev), m.bias.detach().to(dev), 1.0)
    assert relerr(y.cpu(), nhwc(ref.detach())) <= 1e-5


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("count", [1, 12, 19])
def test_running_update_batch_equals_single_updates(dev, count, groups):
    """fcd_instnorm_update_running_batch (one launch for the tracked norms of a generator call; more than 16
    norms go out in several launches) leaves the same bits as one fcd_instnorm_update_running per norm."""
    from fcd_amd import ops
    torch.manual_seed(5)
    n, c = 32, 256
    items, single = [], []
    for _ in range(count):
        rm, rv = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5
        mean, varu = torch.randn(n, c, device=dev), torch.rand(n, c, device=dev) + 0.1
        items.append((rm.clone(), rv.clone(), mean, varu))
        ops.instnorm_update_running(rm, rv, mean, varu, groups=groups)
        single.append((rm, rv))
    ops.instnorm_update_running_batch(items, groups=groups)
    for (rm_b, rv_b, _, _), (rm_s, rv_s) in zip(items, single):
        assert torch.equal(rm_b, rm_s) and torch.equal(rv_b, rv_s)
    with pytest.raises(RuntimeError):
        ops.instnorm_update_running_batch(items[:1] + [(items[0][0], items[0][1], items[0][2][:16], items[0][3][:16])])


@pytest.mark.parametrize("rows,c", [(64, 2048), (65536, 64), (100, 12), (4096, 256)])
def test_colsum(dev, rows, c):
    from fcd_amd import ops
    torch.manual_seed(2)
    x = torch.randn(rows, c)
    out = ops.colsum(rows, x.to(dev), c)
    assert relerr(out, x.double().sum(0)) <= 1e-6
    out2 = ops.colsum(rows, x.to(dev), c, out=out.clone(), accumulate=True)
    assert relerr(out2, 2 * x.double().sum(0)) <= 1e-6


@pytest.mark.parametrize("o,k,pad", [(1, 3, 1), (1, 2, 0), (5, 2, 0)])
def test_discriminator_heads(dev, o, k, pad):
    from fcd_amd import ops
    torch.manual_seed(3)
    n, i, h, w = 4, 64, 2, 2
    x = torch.randn(n, i, h, w, requires_grad=True)
    wt = (torch.randn(o, i, k, k) / (i * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, wt, None, 1, pad)
    gy = torch.randn_like(y)
    y.backward(gy)
    g = ops.ConvGeom(o, i, k, 1, pad)
    wp = ops.pack_weight_fwd(wt.detach().to(dev), g)
    xd = nhwc(x.detach()).to(dev)
    out = ops.head_fwd(xd, wp, o, k, pad)
    assert relerr(out, y) <= 1e-5
    mask = torch.randn(n, i, h, w)
    din = torch.empty_like(xd)
    ops.head_dgrad(gy.to(dev), wp, din, nhwc(mask).to(dev), 0.01, False, k, pad)
    assert relerr(din.cpu(), nhwc(x.grad * torch.where(mask > 0, 1.0, 0.01))) <= 1e-5
    dw = ops.head_wgrad(gy.to(dev), xd, i, k, pad)
    assert relerr(dw, wt.grad) <= 1e-5


def test_losses_forward_backward(dev):
    from fcd_amd import losses
    torch.manual_seed(4)
    a = torch.randn(3, 10, 16, 16)
    b = torch.randn(3, 10, 16, 16)
    b[0, 0, 0, :4] = a[0, 0, 0, :4]            # exact ties: sign(0) = 0 like torch
    ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (3.0 * losses.l1_mean(ad, bd)).backward()
    (3.0 * torch.mean(torch.abs(ar - br))).backward()
    assert relerr(ad.grad, ar.grad) <= 1e-6 and relerr(bd.grad, br.grad) <= 1e-6
    assert abs(float(losses.l1_mean(ad, bd)) - float(torch.mean(torch.abs(a - b)))) <= 1e-6

    x = torch.randn(16, 1) * 3
    t = (torch.rand(16, 1) < 0.5).float()
    xd, xr = x.to(dev).requires_grad_(True), x.clone().requires_grad_(True)
    l = losses.bce_logits_mean(xd, t.to(dev)); l.backward()
    lr = F.binary_cross_entropy_with_logits(xr, t); lr.backward()
    assert abs(float(l) - float(lr)) <= 1e-6 and relerr(xd.grad, xr.grad) <= 1e-6

    s = torch.randn(16, 1, 2, 2)
    sd, sr = s.to(dev).requires_grad_(True), s.clone().requires_grad_(True)
    l = losses.mean_scaled(sd, -1.0); (2 * l).backward()
    lr = -torch.mean(sr); (2 * lr).backward()
    assert abs(float(l) - float(lr)) <= 1e-6 and relerr(sd.grad, sr.grad) <= 1e-6

    g = torch.randn(4, 1000)
    gd, gr = g.to(dev).requires_grad_(True), g.clone().requires_grad_(True)
    l = losses.gp_loss(gd); l.backward()
    lr = torch.mean((torch.sqrt(torch.sum(gr ** 2, dim=1)) - 1) ** 2); lr.backward()
    assert abs(float(l) - float(lr)) <= 1e-5 * abs(float(lr)) and relerr(gd.grad, gr.grad) <= 1e-5


def test_adam_matches_torch(dev):
    from fcd_amd import FusedAdam
    torch.manual_seed(5)
    p0 = torch.randn(1000, 7)
    pr = torch.nn.Parameter(p0.clone())
    pd = torch.nn.Parameter(p0.clone().to(dev))
    o_ref = torch.optim.Adam([pr], 1e-4, (0.5, 0.999))
    o_hip = FusedAdam([pd], 1e-4, (0.5, 0.999))
    for _ in range(3):
        g = torch.randn_like(p0)
        pr.grad = g.clone(); pd.grad = g.clone().to(dev)
        o_ref.step(); o_hip.step()
    assert relerr(pd, pr) <= 1e-6
    assert relerr(pd.detach().cpu() - p0, pr.detach() - p0) <= 1e-4


def test_layout_and_epilogue(dev):
    from fcd_amd import ops
    torch.manual_seed(6)
    x = torch.randn(2, 10, 9, 11)
    x[0, :, :3, :4] = -1.0
    c = torch.tensor([[1.0], [0.0]])
    out = ops.g_prologue(x.to(dev), c.to(dev), 12).cpu()
    assert torch.equal(out[..., :10], nhwc(x)) and bool((out[..., 11] == 0).all())
    assert bool((out[0, ..., 10] == 1).all()) and bool((out[1, ..., 10] == 0).all())
    back = ops.nhwc_to_nchw(out.to(dev), 10).cpu()
    assert torch.equal(back, x)
    delta = torch.randn(2, 9, 11, 12)
    xf = ops.g_epilogue_fwd(delta.to(dev), x.to(dev)).cpu()
    ref = torch.tanh(delta[..., :10].permute(0, 3, 1, 2) + x)
    inv = (x == -1).float()
    ref = ref * (1 - inv) - inv
    assert relerr(xf, ref) <= 1e-6 and bool((xf[x == -1] == -1).all())
    gout = torch.randn_like(x)
    gx = torch.empty_like(x).to(dev)
    gd = ops.g_epilogue_bwd(gout.to(dev), x.to(dev), xf.to(dev), 12, gx).cpu()
    gref = gout * (1 - ref ** 2) * (1 - inv)
    assert relerr(gd[..., :10], nhwc(gref)) <= 1e-5 and float(gd[..., 10:].abs().sum()) == 0
    assert relerr(gx.cpu(), gref) <= 1e-5
    a, b = torch.randn(4, 10, 8, 8), torch.randn(4, 10, 8, 8)
    al = torch.rand(4)
    lr = ops.lerp_rows(a.to(dev), b.to(dev), al.to(dev)).cpu()
    assert relerr(lr, al.view(4, 1, 1, 1) * a + (1 - al.view(4, 1, 1, 1)) * b) <= 1e-6


def test_diffmap_threshold_bit_exact(dev):
    import json, os
    import numpy as np
    from fcd_amd import ops
    torch.manual_seed(7)
    x = torch.rand(3, 10, 16, 16) * 2 - 1
    xf = (x + 0.08 * torch.randn_like(x)).clamp(-1, 1)
    for thr in (0.02, 0.027, 0.02794):
        diff, mask = ops.diffmap_threshold(xf.to(dev), x.to(dev), thr)
        ref = torch.mean(torch.abs(xf - x) / 2, dim=1)
        assert relerr(diff, ref) <= 1e-6
        assert torch.equal(mask.cpu(), (diff.cpu() > thr).to(torch.uint8))     # same fp32 diff -> same mask
    # fixture values at / around fp32(thr): feed them as a 1-band difference (x_fake = x + 2 v)
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "threshold_cases.json")))
    for thr_s, c in cases.items():
        vals = torch.from_numpy(np.array(c["vals_bits"], dtype=np.uint32).view(np.float32).copy())
        x0 = torch.zeros(1, 1, 1, vals.numel())
        xf0 = (2 * vals).view(1, 1, 1, -1)                 # |xf - x| / 2 == vals exactly
        diff, mask = ops.diffmap_threshold(xf0.to(dev), x0.to(dev), float(thr_s))
        assert torch.equal(diff.cpu().flatten(), vals)
        assert mask.cpu().flatten().tolist() == c["torch_gt"]


def _split3_ref(v):
    """fp32 -> three bf16 pieces, round to nearest even at each level (the layout of the *_x3 packs)."""
    h = v.to(torch.bfloat16)
    r = v - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    return torch.stack([h, m, l])


# (O, I, k, s, p): 4x4 shapes take the coalesced pack kernels (thread per (o, i) pair / 64 x 4 x 16 LDS tile),
# the others (and O not a multiple of 64 for the adjoint) the element-per-thread ones
@pytest.mark.parametrize("geom", [(128, 64, 4, 2, 1), (192, 8, 4, 2, 1), (40, 12, 4, 2, 1), (64, 10, 4, 2, 1),
                                  (256, 32, 3, 1, 1), (10, 64, 7, 1, 3)])
def test_weight_packs_match_the_documented_layout(dev, geom):
    """fwd: [o][kh][kw][i]; adj: [phase][i][th][tw][o] with the taps of include/fcd_hip.h (stride 1: flipped;
    4x4 stride 2: kh = 3 - py - 2 th); *_x3: the same elements as three bf16 planes.  Bit-exact."""
    from fcd_amd import ops
    O, I, k, s, p = geom
    g = ops.ConvGeom(O, I, k, s, p)
    torch.manual_seed(9)
    w = torch.randn(O, I, k, k)
    fwd = torch.zeros(O, k, k, g.IP)
    fwd[..., :I] = w.permute(0, 2, 3, 1)
    if s == 1:
        adj = torch.zeros(1, I, k, k, g.OP)
        adj[0, ..., :O] = w.flip(2, 3).permute(1, 2, 3, 0)
    else:
        adj = torch.zeros(4, I, 2, 2, g.OP)
        for ph in range(4):
            py, px = ph >> 1, ph & 1
            for th in range(2):
                for tw in range(2):
                    adj[ph, :, th, tw, :O] = w[:, :, 3 - py - 2 * th, 3 - px - 2 * tw].t()
    wd = w.to(dev)
    assert torch.equal(ops.pack_weight_fwd(wd, g).cpu(), fwd)
    assert torch.equal(ops.pack_weight_adj(wd, g).cpu(), adj)
    assert torch.equal(ops.pack_weight_fwd_x3(wd, g).cpu().view(torch.int16), _split3_ref(fwd).view(torch.int16))
    ax3 = ops.pack_weight_adj_x3(wd, g).cpu()                      # [phase][plane][i][th][tw][o]
    assert torch.equal(ax3.view(torch.int16), _split3_ref(adj).transpose(0, 1).contiguous().view(torch.int16))


@pytest.mark.parametrize("ratio", [100.0, 3000.0])
def test_instnorm_stats_large_mean(dev, ratio):
    """Channels whose |mean| is `ratio` times their standard deviation (fixed_point_gan.py:28,46 put no bound on
    it): mean / variance / rstd from the two-pass kernels AND from the statistics fused into the producing
    convolution's epilogue must still match a float64 evaluation to 1e-5 (the partial sums are taken about a
    per-thread shift, so sum z^2 does not swallow the variance)."""
    from fcd_amd import _lib as L, ops
    torch.manual_seed(7)
    n, h, w, c = 4, 64, 64, 64
    sign = torch.where(torch.arange(c) % 2 == 0, 1.0, -1.0)
    std = 0.5 + torch.rand(c)
    z = torch.randn(n, h, w, c) * std + ratio * std * sign           # NHWC
    z64 = z.double()
    m_ref, v_ref = z64.mean((1, 2)), z64.var((1, 2), unbiased=False)
    zd = z.to(dev)
    mean, rstd, varu = ops.instnorm_stats(zd, want_var=True)
    assert relerr(mean, m_ref) <= 1e-6
    assert float(((rstd.cpu().double() - (v_ref + 1e-5).rsqrt()) / (v_ref + 1e-5).rsqrt()).abs().max()) <= 1e-5
    assert float(((varu.cpu().double() - z64.var((1, 2), unbiased=True)) / v_ref).abs().max()) <= 2e-5
    # fused form: a 1x1 "convolution" with a diagonal weight reproduces z as the conv output
    for prec in (L.PREC_F32, L.PREC_BF16X3):
        g = ops.ConvGeom(c, c, 1, 1, 0)
        d = g.fwd_desc(n, h, w, precision=prec)
        wt = torch.eye(c).view(c, c, 1, 1).contiguous().to(dev)
        wp = ops.pack_weight_fwd_x3(wt, g) if ops.wants_split_weights(d) else ops.pack_weight_fwd(wt, g)
        out, mean_f, rstd_f, varu_f = ops.conv_gather_stats(d, zd, wp, want_var=True)
        o64 = out.cpu().double()
        mo, vo = o64.mean((1, 2)), o64.var((1, 2), unbiased=False)
        assert relerr(out, z) <= 1e-6
        assert relerr(mean_f, mo) <= 1e-6
        assert float(((rstd_f.cpu().double() - (vo + 1e-5).rsqrt()) / (vo + 1e-5).rsqrt()).abs().max()) <= 1e-5, prec


@pytest.mark.parametrize("case", [(32, 64, 0.0, False), (5, 512, 0.2, False), (33, 96, 1.0, False), (32, 64, 0.0, True)])
def test_instnorm_backward_one_pass(dev, case, monkeypatch):
    """InstanceNorm backward of 32 x 32 images (the residual blocks) in one pass -- a workgroup per (image, 32 channels)
    with both operand slices in registers -- against torch in float64, against the three-kernel form (FCD_NO_IN1P=1:
    same sums up to their order), with accumulation into existing affine gradients and with the fused max |dz|."""
    from fcd_amd import ops
    n, c, slope, use_running = case
    torch.manual_seed(3)
    z = (torch.randn(n, c, 32, 32) * 2 + 0.5).double().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(c)).double().requires_grad_(True)
    beta = (0.2 * torch.randn(c)).double().requires_grad_(True)
    gy = torch.randn(n, c, 32, 32).double() * 1e-2
    rm, rv = torch.randn(c).double() * 0.1 + 0.5, torch.rand(c).double() + 3.5
    if use_running:
        y0 = F.instance_norm(z, rm.clone(), rv.clone(), gamma, beta, False, 0.1, 1e-5)
    else:
        y0 = F.instance_norm(z, None, None, gamma, beta, True, 0.1, 1e-5)
    F.leaky_relu(y0, slope).backward(gy)
    zd, gd = nhwc(z.detach().float()).to(dev), nhwc(gy.float()).to(dev)
    gam, bet = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    if use_running:
        mean, rstd = ops.instnorm_stats_from_running(rm.float().to(dev), rv.float().to(dev), n)
    else:
        mean, rstd, _ = ops.instnorm_stats(zd)
    am = torch.zeros(ops.AMAX_SLOTS, device=dev)
    dz, dg, db = ops.instnorm_bwd(gd, None, zd, mean, rstd, gam, slope, use_running, beta=bet, dz_amax=am)
    assert relerr(dz.cpu(), nhwc(z.grad)) <= 2e-5
    assert relerr(dg, gamma.grad) <= 2e-5 and relerr(db, beta.grad) <= 2e-5
    assert float(am.max()) == float(dz.abs().max())
    monkeypatch.setenv("FCD_NO_IN1P", "1")
    dz3, dg3, db3 = ops.instnorm_bwd(gd, None, zd, mean, rstd, gam, slope, use_running, beta=bet)
    monkeypatch.delenv("FCD_NO_IN1P")
    assert relerr(dz, dz3) <= 1e-6 and relerr(dg, dg3) <= 1e-6 and relerr(db, db3) <= 1e-6
    acc_g, acc_b = torch.full_like(dg, 2.0), torch.full_like(db, -1.0)
    ops.instnorm_bwd(gd, None, zd, mean, rstd, gam, slope, use_running, beta=bet, dgamma=acc_g, dbeta=acc_b, accumulate=True)
    assert relerr(acc_g - 2.0, dg) <= 1e-5 and relerr(acc_b + 1.0, db) <= 1e-5


@pytest.mark.parametrize("hw", [32, 20])
def test_instnorm_backward_deferred_affine(dev, hw):
    """fcd_instnorm_bwd_s12 + fcd_instnorm_affine_batch (one launch for several norms, grouped by (N, C)) == the affine
    gradients fcd_instnorm_bwd accumulates itself, bit for bit, for the one-pass (32 x 32) and the three-kernel form."""
    from fcd_amd import ops
    torch.manual_seed(11)
    items, want = [], []
    for (n, c, slope) in [(8, 256, 0.0), (8, 256, 1.0), (8, 64, 0.2), (3, 64, 0.0), (8, 256, 0.0)]:
        z = torch.randn(n, hw, hw, c, device=dev) * 2 + 0.4
        gy = torch.randn(n, hw, hw, c, device=dev) * 1e-2
        gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.2
        mean, rstd, _ = ops.instnorm_stats(z)
        g0, b0 = torch.randn(c, device=dev), torch.randn(c, device=dev)
        rg, rb = g0.clone(), b0.clone()
        am0, am1 = torch.zeros(ops.AMAX_SLOTS, device=dev), torch.zeros(ops.AMAX_SLOTS, device=dev)
        dz0, _, _ = ops.instnorm_bwd(gy, None, z, mean, rstd, gamma, slope, False, beta=beta, dgamma=rg, dbeta=rb,
                                     accumulate=True, dz_amax=am0)
        dz1, s12 = ops.instnorm_bwd_deferred(gy, z, mean, rstd, gamma, slope, False, beta, dz_amax=am1)
        assert torch.equal(dz0, dz1) and torch.equal(am0, am1)
        dg, db = g0.clone(), b0.clone()
        items.append((s12, dg, db))
        want.append((rg, rb))
    items[-1] = (items[-1][0], items[-1][1], None)        # a norm whose bias gradient is not wanted
    ops.instnorm_affine_batch(items, accumulate=True)
    for (s12, dg, db), (rg, rb) in zip(items, want):
        assert torch.equal(dg, rg)
        assert db is None or torch.equal(db, rb)
    # overwrite instead of accumulate
    dg, db = torch.full((256,), 7.0, device=dev), torch.full((256,), 7.0, device=dev)
    ops.instnorm_affine_batch([(items[0][0], dg, db)], accumulate=False)
    assert torch.equal(dg, items[0][0][:, :, 1].double().sum(0).float()) or relerr(dg, items[0][0][:, :, 1].double().sum(0)) <= 1e-6
    with pytest.raises(RuntimeError):
        ops.instnorm_affine_batch([(items[0][0], None, None)], accumulate=True)


def test_absmax_batch(dev):
    """fcd_absmax_batch: rows overwritten (stale contents do not leak), equal to per-tensor maxima; more tensors than one
    launch takes; refusal of unaligned lengths."""
    from fcd_amd import ops
    torch.manual_seed(12)
    ts = [torch.randn(n, device=dev) * (i + 1) for i, n in enumerate([4, 1024, 37632, 589824, 2 ** 21 + 8] + [64] * 31)]
    out = torch.full((len(ts), ops.AMAX_SLOTS), 1e9, device=dev)
    ops.absmax_batch(ts, out)
    for i, t in enumerate(ts):
        assert float(out[i].max()) == float(t.abs().max()), i
        assert float(out[i].min()) >= 0.0
    ts[1].zero_()
    ops.absmax_batch(ts, out)
    assert float(out[1].max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.absmax_batch([torch.randn(6, device=dev)], torch.empty((1, ops.AMAX_SLOTS), device=dev))
    with pytest.raises(ValueError):
        ops.absmax_batch(ts, torch.empty((2, ops.AMAX_SLOTS), device=dev))
