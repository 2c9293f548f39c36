"""HIP implicit-GEMM convolutions (through the C ABI) against torch's CPU convolutions.

fp32 MFMA path: rel-err <= 1e-5 per tensor (the bar of the full step is 1e-4);
bf16 MFMA path: rel-err <= 1e-2 (BASELINE.json north_star tolerances).
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

pytestmark = pytest.mark.gpu

TOL = {0: 1e-5, 1: 1e-2, 2: 1e-5}    # 2 = bf16x3 split: fp32-faithful, same bar as exact fp32

# (O, I, k, s, p, N, H, W)
GEOMS = [
    (64, 11, 7, 1, 3, 2, 20, 24),      # G first conv (Cin 11 -> 12 physical)
    (128, 64, 4, 2, 1, 2, 16, 16),     # down-sampling
    (256, 256, 3, 1, 1, 3, 8, 8),      # residual block conv
    (10, 64, 7, 1, 3, 2, 12, 12),      # G last conv (Cout 10)
    (96, 40, 3, 1, 1, 1, 9, 7),        # ragged sizes
    (512, 256, 4, 2, 1, 4, 4, 4),      # D deep layer: tiny M, split-K
    (16, 8, 4, 2, 1, 2, 32, 32),       # tiny channels
]


def to_nhwc(t, cp):
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cp)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out.contiguous()


def pack_for(d, w, g, form):
    """The weight operand descriptor ``d`` wants (three bf16 planes for BF16X3 on 128x128 tiles)."""
    from fcd_amd import ops
    x3 = ops.wants_split_weights(d)
    if form == "fwd":
        return ops.pack_weight_fwd_x3(w, g) if x3 else ops.pack_weight_fwd(w, g)
    return ops.pack_weight_adj_x3(w, g) if x3 else ops.pack_weight_adj(w, g)


def from_nhwc(t, c):
    return t[..., :c].permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("geom", GEOMS)
def test_conv_forward(dev, geom, prec):
    from fcd_amd import ops
    O, I, k, s, p, N, H, W = geom
    g = ops.ConvGeom(O, I, k, s, p)
    torch.manual_seed(1)
    x = torch.randn(N, I, H, W)
    w = torch.randn(O, I, k, k) / (I * k * k) ** 0.5
    b = torch.randn(O)
    ref = F.leaky_relu(F.conv2d(x, w, b, s, p), 0.01)
    xd = to_nhwc(x, g.IP).to(dev)
    d = g.fwd_desc(N, H, W, precision=prec, act_slope=0.01)
    wf = pack_for(d, w.to(dev), g, "fwd")
    out = ops.conv_gather(d, xd, wf, bias=b.to(dev))
    torch.cuda.synchronize()
    assert relerr(from_nhwc(out.cpu(), O), ref) <= TOL[prec]
    assert float(out[..., O:].abs().sum()) == 0.0   # pad channels stay zero
    # explicit split-K gives the same answer
    d2 = g.fwd_desc(N, H, W, precision=prec, act_slope=0.01)
    d2.splitk = 3
    out2 = ops.conv_gather(d2, xd, pack_for(d2, w.to(dev), g, "fwd"), bias=b.to(dev))
    assert relerr(from_nhwc(out2.cpu(), O), ref) <= TOL[prec]


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("geom", GEOMS)
def test_conv_adjoint(dev, geom, prec):
    """Data gradient of the conv == ConvTranspose2d forward with the same weight."""
    from fcd_amd import ops
    O, I, k, s, p, N, H, W = geom
    g = ops.ConvGeom(O, I, k, s, p)
    sh, sw = g.small_hw(H, W)
    bh, bw = g.big_hw(sh, sw)
    torch.manual_seed(2)
    gy = torch.randn(N, O, sh, sw)
    w = torch.randn(O, I, k, k) / (O * k * k) ** 0.5
    ref = F.conv_transpose2d(gy, w, None, s, p)
    assert ref.shape[2:] == (bh, bw)
    mask = torch.randn(N, I, bh, bw)
    res = torch.randn(N, I, bh, bw)
    ref2 = ref * torch.where(mask > 0, 1.0, 0.01) + res
    d = g.adj_desc(N, sh, sw, precision=prec, mask_slope=0.01)
    wa = pack_for(d, w.to(dev), g, "adj")
    out = ops.conv_gather(d, to_nhwc(gy, g.OP).to(dev), wa)
    assert relerr(from_nhwc(out.cpu(), I), ref) <= TOL[prec]
    out2 = ops.conv_gather(d, to_nhwc(gy, g.OP).to(dev), wa, mask=to_nhwc(mask, g.IP).to(dev),
                           residual=to_nhwc(res, g.IP).to(dev))
    assert relerr(from_nhwc(out2.cpu(), I), ref2) <= TOL[prec]


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("geom", GEOMS)
def test_conv_wgrad(dev, geom, prec):
    from fcd_amd import ops
    O, I, k, s, p, N, H, W = geom
    g = ops.ConvGeom(O, I, k, s, p)
    sh, sw = g.small_hw(H, W)
    torch.manual_seed(3)
    x = torch.randn(N, I, H, W)
    gy = torch.randn(N, O, sh, sw)
    w = torch.zeros(O, I, k, k, requires_grad=True)
    (F.conv2d(x, w, None, s, p) * gy).sum().backward()
    ref = w.grad
    d = g.wgrad_desc(N, H, W, precision=prec)
    dw = ops.conv_wgrad(d, to_nhwc(gy, g.OP).to(dev), to_nhwc(x, g.IP).to(dev))
    assert relerr(dw.cpu(), ref) <= TOL[prec]
    d.accumulate = 1
    d.splitk = 2
    dw2 = ops.conv_wgrad(d, to_nhwc(gy, g.OP).to(dev), to_nhwc(x, g.IP).to(dev), dw.clone())
    assert relerr(dw2.cpu(), 2 * ref) <= TOL[prec]


# (O, I, k, s, p, N, H, W, form): shapes where the statistics ride in the epilogue (H*W a multiple of the
# wave's rows, no split-K) and shapes that fall back to the two-pass path
STAT_GEOMS = [
    (256, 256, 3, 1, 1, 4, 32, 32, "fwd"),     # residual-block conv: fused, 128x128 tiles
    (64, 11, 7, 1, 3, 2, 32, 32, "fwd"),       # first conv: fused, 128x64 tiles
    (128, 64, 4, 2, 1, 2, 32, 32, "fwd"),      # down-sampling
    (128, 64, 4, 2, 1, 2, 16, 16, "adj"),      # ConvTranspose2d (four phases)
    (96, 40, 3, 1, 1, 1, 9, 7, "fwd"),         # ragged: fallback
    (512, 256, 4, 2, 1, 4, 4, 4, "fwd"),       # split-K: fallback
]


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("geom", STAT_GEOMS)
def test_conv_with_fused_instnorm_statistics(dev, geom, prec):
    """fcd_conv_gather_stats == fcd_conv_gather followed by fcd_instnorm_stats."""
    from fcd_amd import ops
    O, I, k, s, p, N, H, W, form = geom
    g = ops.ConvGeom(O, I, k, s, p)
    torch.manual_seed(4)
    w = torch.randn(O, I, k, k) / (I * k * k) ** 0.5
    if form == "fwd":
        x = to_nhwc(torch.randn(N, I, H, W) + 0.3, g.IP).to(dev)
        d = g.fwd_desc(N, H, W, precision=prec)
        wp = pack_for(d, w.to(dev), g, "fwd")
    else:
        x = to_nhwc(torch.randn(N, O, H, W) + 0.3, g.OP).to(dev)
        d = g.adj_desc(N, H, W, precision=prec)
        wp = pack_for(d, w.to(dev), g, "adj")
    z0 = ops.conv_gather(d, x, wp)
    m0, r0, v0 = ops.instnorm_stats(z0, want_var=True)
    z1, m1, r1, v1 = ops.conv_gather_stats(d, x, wp, want_var=True)
    assert torch.equal(z0, z1)
    assert relerr(m1, m0) <= 1e-5 and relerr(r1, r0) <= 1e-5 and relerr(v1, v0) <= 1e-5
    zc = z1.cpu().double()
    assert relerr(m1, zc.mean((1, 2))) <= 1e-5
    assert relerr(v1, zc.var((1, 2), unbiased=True)) <= 1e-5


APPLY_GEOMS = [
    # O, I, k, s, p, N, H, W, form, act_slope, residual
    (256, 256, 3, 1, 1, 8, 32, 32, "fwd", 0.0, False),     # residual block, first half: one-pass kernel
    (256, 256, 3, 1, 1, 8, 32, 32, "fwd", 1.0, True),      # second half: no activation, + residual
    (64, 64, 3, 1, 1, 33, 32, 32, "fwd", 0.2, True),       # odd image count, LeakyReLU
    (256, 128, 4, 2, 1, 8, 64, 64, "fwd", 0.0, False),     # strided layer onto 32 x 32
    (64, 128, 4, 2, 1, 4, 32, 32, "adj", 0.0, False),      # 64 x 64 output: finalize + apply
    (256, 256, 3, 1, 1, 4, 32, 32, "fwd", "silu", False),  # smooth activation: finalize + apply
    (96, 40, 3, 1, 1, 1, 9, 7, "fwd", 0.0, True),          # ragged: two-pass statistics + apply
    (32, 32, 3, 1, 1, 1, 32, 32, "fwd", 0.0, False),       # too few workgroups for the one-pass form
]


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("geom", APPLY_GEOMS)
def test_conv_stats_apply_in_one_call(dev, geom, prec, monkeypatch):
    """fcd_conv_gather_stats_apply == fcd_conv_gather_stats followed by fcd_instnorm_apply_amax, bit for bit (the one-pass
    kernel finishes the statistics in finalize_kernel's order), with FCD_NO_IN1F=1 as well, and y against torch."""
    from fcd_amd import ops
    O, I, k, s, p, N, H, W, form, slope, with_res = geom
    if slope == "silu":
        slope = ops.ACT_SILU
    g = ops.ConvGeom(O, I, k, s, p)
    torch.manual_seed(6)
    w = torch.randn(O, I, k, k) / (I * k * k) ** 0.5
    if form == "fwd":
        x = to_nhwc(torch.randn(N, I, H, W) + 0.3, g.IP).to(dev)
        d = g.fwd_desc(N, H, W, precision=prec)
        wp = pack_for(d, w.to(dev), g, "fwd")
        cout = g.OP
    else:
        x = to_nhwc(torch.randn(N, O, H, W) + 0.3, g.OP).to(dev)
        d = g.adj_desc(N, H, W, precision=prec)
        wp = pack_for(d, w.to(dev), g, "adj")
        cout = g.IP
    gamma = (1 + 0.2 * torch.randn(cout)).to(dev)
    beta = (0.2 * torch.randn(cout)).to(dev)
    z0, m0, r0, v0 = ops.conv_gather_stats(d, x, wp, want_var=True)
    res = torch.randn_like(z0) if with_res else None
    am0 = torch.zeros(ops.AMAX_SLOTS, device=dev)
    y0 = ops.instnorm_apply(z0, m0, r0, gamma, beta, slope, residual=res, amax=am0)
    for off in (False, True):
        if off:
            monkeypatch.setenv("FCD_NO_IN1F", "1")
        am1 = torch.zeros(ops.AMAX_SLOTS, device=dev)
        z1, m1, r1, v1, y1 = ops.conv_gather_stats(d, x, wp, want_var=True, apply=(gamma, beta, slope, res, am1))
        assert torch.equal(z0, z1) and torch.equal(m0, m1) and torch.equal(r0, r1) and torch.equal(v0, v1)
        assert torch.equal(y0, y1)
        assert float(am1.max()) == float(y1.abs().max()) == float(am0.max())
        # without the unbiased variance and the maximum
        _, m2, _, v2, y2 = ops.conv_gather_stats(d, x, wp, apply=(gamma, beta, slope, None, None))
        assert v2 is None and torch.equal(m2, m0)
        if res is None:
            assert torch.equal(y2, y0)
    zc = z0.cpu().double()
    mu, var = zc.mean((1, 2), keepdim=True), zc.var((1, 2), unbiased=False, keepdim=True)
    ref = (zc - mu) / (var + 1e-5).sqrt() * gamma.cpu().double() + beta.cpu().double()
    ref = F.silu(ref) if slope == ops.ACT_SILU else F.leaky_relu(ref, slope)
    if res is not None:
        ref = ref + res.cpu().double()
    assert relerr(y0.cpu(), ref) <= 1e-5


# 7x7 / <= 16 output channels / input channels a multiple of 16: the halo weight-gradient kernel
# (input rows loaded once for all 49 taps).  Ragged widths (column strips with out-of-range halo),
# heights that do not divide into the row ranges, 3 / 10 / 16 output channels, accumulation.
HALO_GEOMS = [
    (10, 64, 2, 128, 128),
    (10, 32, 3, 40, 100),
    (16, 16, 1, 70, 64),
    (3, 48, 2, 33, 130),
    (12, 16, 5, 9, 7),
]


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("geom", HALO_GEOMS)
def test_wgrad_7x7_halo(dev, geom, prec):
    from fcd_amd import ops
    O, I, N, H, W = geom
    g = ops.ConvGeom(O, I, 7, 1, 3)
    torch.manual_seed(5)
    x = torch.randn(N, I, H, W)
    gy = torch.randn(N, O, H, W)
    w = torch.zeros(O, I, 7, 7, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, None, 1, 3) * gy.double()).sum().backward()
    ref = w.grad
    d = g.wgrad_desc(N, H, W, precision=prec)
    gyd, xd = to_nhwc(gy, g.OP).to(dev), to_nhwc(x, g.IP).to(dev)
    dw = ops.conv_wgrad(d, gyd, xd)
    from fcd_amd import _lib as L
    assert L.load().fcd_last_conv_kernel() == (b"wgrad7_halo_x3_kernel" if prec == 2 else b"wgrad7_halo_kernel")
    assert relerr(dw.cpu(), ref) <= (1e-6 if prec == 2 else 1e-5)     # the split-once BF16X3 kernel is fp32-faithful
    d.accumulate = 1
    dw2 = ops.conv_wgrad(d, gyd, xd, dw.clone())
    assert relerr(dw2.cpu(), 2 * ref) <= 1e-5
    # the generic 16-row kernel (explicit split) computes the same thing
    d2 = g.wgrad_desc(N, H, W, precision=prec)
    d2.splitk = 3
    dw3 = ops.conv_wgrad(d2, gyd, xd)
    assert relerr(dw3.cpu(), dw.cpu()) <= 1e-5


# 256-row BF16X3 gather tiles (csrc/conv_igemm.hip gather_big_tile: chosen when the grid still has >= 256
# workgroups).  Samples are independent, so a launch on all samples (256-row tiles) must give the same bits
# as launches on slices small enough to take the 128-row tiles -- and both must meet the fp32 bar against
# torch.  The slices are sized to stay above the split-K threshold (>= 192 tiles of 128 rows) and below
# the 256-row one.  (O, I, k, s, p, N, H, W, form, slice)
BIG_TILE_GEOMS = [
    (256, 256, 3, 1, 1, 32, 32, 32, "fwd", 16),     # residual-block conv: 256x128 tile, 256 workgroups
    (256, 256, 3, 1, 1, 40, 30, 30, "fwd", 20),     # ragged: M = 36000 is not a multiple of 256
    (256, 128, 4, 2, 1, 40, 60, 60, "fwd", 20),     # strided down-sampling, ragged
    (128, 64, 4, 2, 1, 18, 62, 62, "adj", 2),       # 128 -> 64 up-sampling adjoint: 256x64 tile, four phases, ragged
    (64, 64, 3, 1, 1, 18, 62, 62, "fwd", 9),        # 64 output channels, one phase: 256x64 tile
]


@pytest.mark.parametrize("geom", BIG_TILE_GEOMS)
def test_big_tile_equals_small_tile(dev, geom):
    from fcd_amd import ops
    O, I, k, s, p, N, H, W, form, step = geom
    g = ops.ConvGeom(O, I, k, s, p)
    torch.manual_seed(7)
    w = torch.randn(O, I, k, k) / (I * k * k) ** 0.5
    b = torch.randn(O if form == "fwd" else I)
    if form == "fwd":
        xc = torch.randn(N, I, H, W)
        ref = F.leaky_relu(F.conv2d(xc, w, b, s, p), 0.01)
        x = to_nhwc(xc, g.IP).to(dev)
        desc = lambda n: g.fwd_desc(n, H, W, precision=2, act_slope=0.01)
        cout = O
    else:
        xc = torch.randn(N, O, H, W)
        ref = F.leaky_relu(F.conv_transpose2d(xc, w, b, s, p), 0.01)
        x = to_nhwc(xc, g.OP).to(dev)
        desc = lambda n: g.adj_desc(n, H, W, precision=2, act_slope=0.01)
        cout = I
    wp = pack_for(desc(N), w.to(dev), g, form)
    full = ops.conv_gather(desc(N), x, wp, bias=b.to(dev))
    assert relerr(from_nhwc(full.cpu(), cout), ref) <= TOL[2]
    for i in range(0, N, step):
        part = ops.conv_gather(desc(step), x[i:i + step].contiguous(), wp, bias=b.to(dev))
        assert torch.equal(part, full[i:i + step]), i
    if form == "fwd" and s == 1:
        # fused InstanceNorm statistics on the 256-row tiles (64 / 32 rows per wave)
        d = g.fwd_desc(N, H, W, precision=2)
        z0 = ops.conv_gather(d, x, wp)
        z1, m1, r1, v1 = ops.conv_gather_stats(d, x, wp, want_var=True)
        assert torch.equal(z0, z1)
        zc = z1.cpu().double()
        assert relerr(m1, zc.mean((1, 2))) <= 1e-5
        assert relerr(v1, zc.var((1, 2), unbiased=True)) <= 1e-5


# 7x7 / 64 input channels / <= 16 output channels: the forward halo kernel (input rows loaded once for all
# 49 taps).  Ragged widths (strips with out-of-range halo), heights that do not divide into the row ranges,
# 3 / 10 / 16 output channels, with and without bias.  (O, N, H, W, bias)
HALO_FWD_GEOMS = [
    (10, 2, 128, 128, False),
    (10, 3, 21, 45, True),
    (3, 1, 40, 33, True),
    (16, 2, 9, 70, False),
    (12, 5, 64, 64, True),
]


@pytest.mark.parametrize("prec", [0, 2])
@pytest.mark.parametrize("geom", HALO_FWD_GEOMS)
def test_conv_7x7_halo_forward(dev, geom, prec):
    from fcd_amd import ops
    O, N, H, W, with_bias = geom
    g = ops.ConvGeom(O, 64, 7, 1, 3)
    torch.manual_seed(11)
    x = torch.randn(N, 64, H, W)
    w = torch.randn(O, 64, 7, 7) / 56
    b = torch.randn(O) if with_bias else None
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), None if b is None else b.double(), 1, 3), 0.2)
    xd = to_nhwc(x, g.IP).to(dev)
    d = g.fwd_desc(N, H, W, precision=prec, act_slope=0.2)
    d.splitk = 1           # (small problems would otherwise split K and take the generic kernel)
    out = ops.conv_gather(d, xd, pack_for(d, w.to(dev), g, "fwd"), bias=None if b is None else b.to(dev))
    assert relerr(from_nhwc(out.cpu(), O).double(), ref) <= TOL[prec]
    assert float(out[..., O:].abs().max()) == 0.0 if g.OP > O else True     # padded channels stay zero
    from fcd_amd import _lib as L
    assert L.load().fcd_last_conv_kernel() == (b"conv7_halo_x3_kernel" if prec == 2 else b"conv7_halo_kernel")
    if prec == 2:
        # the split-once BF16X3 kernel is fp32-faithful: as close to float64 as the exact-fp32 halo kernel
        assert relerr(from_nhwc(out.cpu(), O).double(), ref) <= 1e-6


@pytest.mark.parametrize("shape", [(256, 256, 3, 1, 1, 4, 32, 32), (300, 36, 3, 1, 1, 3, 17, 13), (128, 64, 4, 2, 1, 2, 64, 64),
                                   (64, 12, 7, 1, 3, 2, 40, 24), (40, 8, 4, 2, 1, 5, 10, 6), (512, 256, 4, 2, 1, 3, 16, 16)])
def test_wgrad_split_once_equals_in_register_split(dev, shape, monkeypatch):
    """FCD_PREC_BF16X3 weight gradient: the split-once kernel (operands split one time per workgroup, bf16
    planes in LDS, transposed reads; conv_wgrad_x3p_kernel) forms the same products in the same k slots and
    accumulators as the kernel that splits every fragment in registers (FCD_WGRAD_OLD=1): identical bits,
    and both within 1e-5 of torch's weight gradient."""
    import torch.nn.functional as F
    from fcd_amd import ops
    O, I, k, s, pd, n, h, w = shape
    g = ops.ConvGeom(O, I, k, s, pd)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, I, h, w, generator=gen)
    sh, sw = g.small_hw(h, w)
    gy = torch.randn(n, O, sh, sw, generator=gen)
    xd = torch.zeros(n, h, w, g.IP)
    xd[..., :I] = x.permute(0, 2, 3, 1)
    gd = torch.zeros(n, sh, sw, g.OP)
    gd[..., :O] = gy.permute(0, 2, 3, 1)
    xd, gd = xd.to(dev), gd.to(dev)
    wd = g.wgrad_desc(n, h, w, precision=2)
    monkeypatch.setenv("FCD_WGRAD_OLD", "1")
    old = ops.conv_wgrad(wd, gd, xd).clone()
    monkeypatch.delenv("FCD_WGRAD_OLD")
    new = ops.conv_wgrad(wd, gd, xd)
    assert torch.equal(old, new)
    ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, k, k), gy.double(), stride=s, padding=pd)
    assert relerr(new, ref) <= 1e-5
    acc = torch.full_like(new, 0.5)
    wd2 = g.wgrad_desc(n, h, w, precision=2, accumulate=True)
    ops.conv_wgrad(wd2, gd, xd, dw=acc)
    assert relerr(acc - 0.5, ref) <= 1e-5


@pytest.mark.parametrize("shape", [(2, 40, 24, 11), (1, 9, 130, 10), (3, 64, 64, 11), (4, 128, 128, 11)])
def test_conv7_c12_halo_x3(dev, shape, monkeypatch):
    """The 7x7, 12 -> 64 channel convolution of the generator's first layer (fixed_point_gan.py:45) and of the last
    layer's data gradient under FCD_PREC_BF16X3: the LDS-halo kernel (input rows loaded and split once, one wave
    per tap row, fixed-order sum of the 7 partial tiles) against torch's conv2d in float64 and against the generic
    gather kernel (FCD_NO_HALO12=1), with and without the fused InstanceNorm statistics, ragged widths and heights,
    and as the adjoint (data gradient) of the 64 -> 10 convolution."""
    from fcd_amd import _lib as L, ops
    n, h, w, cin = shape
    g = ops.ConvGeom(64, cin, 7, 1, 3)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(64, cin, 7, 7, generator=gen) / 23.0
    xd = torch.zeros(n, h, w, g.IP)
    xd[..., :cin] = x.permute(0, 2, 3, 1)
    xd = xd.to(dev)
    d = g.fwd_desc(n, h, w, precision=2)
    wp = ops.pack_weight_fwd(wt.to(dev), g)
    ref = F.conv2d(x.double(), wt.double(), padding=3).permute(0, 2, 3, 1)
    out, mean, rstd, varu = ops.conv_gather_stats(d, xd, wp, want_var=True)
    assert L.load().fcd_last_conv_kernel() == b"conv7_c12_halo_x3_kernel"
    assert relerr(out, ref) <= 1e-6
    m_ref, v_ref = ref.mean((1, 2)), ref.var((1, 2), unbiased=False)
    assert relerr(mean, m_ref) <= 1e-5
    assert float(((rstd.cpu().double() - (v_ref + 1e-5).rsqrt()) / (v_ref + 1e-5).rsqrt()).abs().max()) <= 1e-5
    assert relerr(varu, ref.var((1, 2), unbiased=True)) <= 1e-5
    plain = ops.conv_gather(d, xd, wp)
    assert torch.equal(plain, out)                      # the statistics do not change the convolution
    monkeypatch.setenv("FCD_NO_HALO12", "1")
    generic = ops.conv_gather(d, xd, wp)
    assert L.load().fcd_last_conv_kernel() != b"conv7_c12_halo_x3_kernel"
    monkeypatch.delenv("FCD_NO_HALO12")
    assert relerr(out, generic) <= 1e-6
    # epilogue operands the halo kernel does not implement fall back to the generic kernel
    bias = torch.randn(64, generator=gen).to(dev)
    withb = ops.conv_gather(d, xd, wp, bias=bias)
    assert L.load().fcd_last_conv_kernel() != b"conv7_c12_halo_x3_kernel"
    assert relerr(withb, ref + bias.cpu().double()) <= 1e-6
    # adjoint use: data gradient of Conv2d(64 -> cout, 7x7) with cout = 10 (12 physical channels)
    cout = 10
    g2 = ops.ConvGeom(cout, 64, 7, 1, 3)
    w2 = torch.randn(cout, 64, 7, 7, generator=gen) / 56.0
    gy = torch.randn(n, cout, h, w, generator=gen)
    gyd = torch.zeros(n, h, w, g2.OP)
    gyd[..., :cout] = gy.permute(0, 2, 3, 1)
    d2 = g2.adj_desc(n, h, w, precision=2)
    gx = ops.conv_gather(d2, gyd.to(dev), ops.pack_weight_adj(w2.to(dev), g2))
    assert L.load().fcd_last_conv_kernel() == b"conv7_c12_halo_x3_kernel"
    gx_ref = torch.nn.grad.conv2d_input((n, 64, h, w), w2.double(), gy.double(), padding=3).permute(0, 2, 3, 1)
    assert relerr(gx, gx_ref) <= 1e-6


@pytest.mark.parametrize("shape", [(64, 11, 2, 40, 24), (64, 10, 1, 9, 130), (40, 12, 3, 64, 64), (64, 11, 4, 128, 128), (17, 11, 2, 33, 70)])
def test_wgrad7_c12_halo_x3(dev, shape, monkeypatch):
    """Weight gradient of the generator's first layer (7x7, 11 -> 64, fixed_point_gan.py:45) under FCD_PREC_BF16X3: the
    halo kernel (rows split once into bf16 planes in LDS, transposed reads along the pixel axis, the 7 taps of a tap
    row as 84 consecutive values of a ring row) against torch in float64 and against the generic split-once kernel
    (FCD_NO_WGRAD12=1); ragged widths / heights, 17 / 40 / 64 output channels, accumulation."""
    from fcd_amd import _lib as L, ops
    O, I, n, h, w = shape
    g = ops.ConvGeom(O, I, 7, 1, 3)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(n, I, h, w, generator=gen)
    gy = torch.randn(n, O, h, w, generator=gen)
    xd, gd = to_nhwc(x, g.IP).to(dev), to_nhwc(gy, g.OP).to(dev)
    ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, 7, 7), gy.double(), stride=1, padding=3)
    d = g.wgrad_desc(n, h, w, precision=2)
    dw = ops.conv_wgrad(d, gd, xd)
    assert L.load().fcd_last_conv_kernel() == b"wgrad7_c12_halo_x3_kernel"
    assert relerr(dw.cpu(), ref) <= 1e-6
    monkeypatch.setenv("FCD_NO_WGRAD12", "1")
    generic = ops.conv_wgrad(d, gd, xd)
    assert L.load().fcd_last_conv_kernel() != b"wgrad7_c12_halo_x3_kernel"
    monkeypatch.delenv("FCD_NO_WGRAD12")
    assert relerr(dw, generic) <= 1e-6
    acc = torch.full_like(dw, 0.25)
    ops.conv_wgrad(g.wgrad_desc(n, h, w, precision=2, accumulate=True), gd, xd, dw=acc)
    assert relerr(acc.cpu() - 0.25, ref) <= 1e-5


# (O, I, N, H, W, form): 3x3, stride 1, pad 1 with image width 32 / 64 and enough tiles for no split-K
AHALO_GEOMS = [
    (256, 256, 32, 32, 32, "fwd"),      # residual-block conv, paired G step: 256-row tiles
    (256, 256, 16, 32, 32, "adj"),      # its data gradient at 16 samples: 128-row tiles
    (128, 96, 9, 64, 64, "fwd"),        # 64-wide images, 96 input channels (3 slices), tiles of 2 / 4 image rows
    (64, 192, 25, 16, 32, "adj"),       # ragged output-channel tile (192 = 128 + 64), short images
]


@pytest.mark.parametrize("geom", AHALO_GEOMS)
def test_gather_a_halo_equals_per_tap_gather(dev, geom, monkeypatch):
    """FCD_PREC_BF16X3 3x3 convolutions: the A-halo form (the tile's halo region split once per 32-channel slice into
    bf16 planes in LDS, nine taps read at shifted pixel addresses; template flag AH) forms the same products in the
    same order as the per-tap gather form (FCD_NO_HALO3=1): identical bits, with bias / activation, mask / residual
    epilogues and the fused InstanceNorm statistics; and within 1e-5 of torch."""
    from fcd_amd import _lib as L, ops
    O, I, N, H, W, form = geom
    g = ops.ConvGeom(O, I, 3, 1, 1)
    gen = torch.Generator().manual_seed(9)
    wt = torch.randn(O, I, 3, 3, generator=gen) / (3 * I ** 0.5)
    if form == "fwd":
        x = torch.randn(N, I, H, W, generator=gen)
        b = torch.randn(O, generator=gen)
        d = g.fwd_desc(N, H, W, precision=2, act_slope=0.01)
        wp = ops.pack_weight_fwd_x3(wt.to(dev), g)
        xd = to_nhwc(x, g.IP).to(dev)
        run = lambda: ops.conv_gather(d, xd, wp, bias=b.to(dev))
        ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), 1, 1), 0.01)
        cref = O
        d0 = g.fwd_desc(N, H, W, precision=2)
        stats = lambda: ops.conv_gather_stats(d0, xd, wp, want_var=True)
    else:
        gy = torch.randn(N, O, H, W, generator=gen)
        mask = torch.randn(N, I, H, W, generator=gen)
        res = torch.randn(N, I, H, W, generator=gen)
        d = g.adj_desc(N, H, W, precision=2, mask_slope=0.01)
        wp = ops.pack_weight_adj_x3(wt.to(dev), g)
        gd, md, rd = to_nhwc(gy, g.OP).to(dev), to_nhwc(mask, g.IP).to(dev), to_nhwc(res, g.IP).to(dev)
        run = lambda: ops.conv_gather(d, gd, wp, mask=md, residual=rd)
        ref = F.conv_transpose2d(gy.double(), wt.double(), None, 1, 1) * torch.where(mask > 0, 1.0, 0.01) + res
        cref = I
        stats = None
    new = run().clone()
    assert L.load().fcd_last_conv_kernel().endswith(b"true, true, 3>")
    st_new = [t.clone() for t in stats()] if stats else None
    monkeypatch.setenv("FCD_NO_HALO3", "1")
    old = run().clone()
    assert not L.load().fcd_last_conv_kernel().endswith(b"true, true, 3>")
    st_old = [t.clone() for t in stats()] if stats else None
    monkeypatch.delenv("FCD_NO_HALO3")
    assert torch.equal(new, old)
    if stats:
        for a, b2 in zip(st_new, st_old):
            assert torch.equal(a, b2)
    assert relerr(from_nhwc(new.cpu(), cref).double(), ref) <= 1e-5


@pytest.mark.parametrize("shape", [(256, 256, 16, 4), (128, 192, 40, 5), (64, 64, 200, 2), (254, 128, 30, 3), (256, 256, 32, 32)])
def test_wgrad3_halo_x3(dev, shape, monkeypatch):
    """Weight gradient of the residual blocks' 3x3 convolutions (fixed_point_gan.py:26-31) on 32-pixel-wide images under
    FCD_PREC_BF16X3: the halo kernel (a workgroup owns 64 x 64 channels and all nine taps; rows split once into bf16
    planes, taps = ring rows and pixel shifts of transposed reads) against torch in float64 and against the generic
    split-once kernel (FCD_NO_WGRAD3=1); several images per split, ragged output channels, accumulation."""
    from fcd_amd import _lib as L, ops
    O, I, n, h = shape
    g = ops.ConvGeom(O, I, 3, 1, 1)
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(n, I, h, 32, generator=gen)
    gy = torch.randn(n, O, h, 32, generator=gen)
    xd, gd = to_nhwc(x, g.IP).to(dev), to_nhwc(gy, g.OP).to(dev)
    d = g.wgrad_desc(n, h, 32, precision=2)
    dw = ops.conv_wgrad(d, gd, xd)
    assert L.load().fcd_last_conv_kernel() == b"wgrad3_halo_x3_kernel"
    monkeypatch.setenv("FCD_NO_WGRAD3", "1")
    generic = ops.conv_wgrad(d, gd, xd)
    assert L.load().fcd_last_conv_kernel() != b"wgrad3_halo_x3_kernel"
    monkeypatch.delenv("FCD_NO_WGRAD3")
    assert relerr(dw, generic) <= 2e-6
    if n * h <= 200:
        ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, 3, 3), gy.double(), stride=1, padding=1)
        assert relerr(dw.cpu(), ref) <= 1e-6
        acc = torch.full_like(dw, 0.25)
        ops.conv_wgrad(g.wgrad_desc(n, h, 32, precision=2, accumulate=True), gd, xd, dw=acc)
        assert relerr(acc.cpu() - 0.25, ref) <= 1e-5


# The scaled-fp16 family: (precision code, bar against float64, template suffix of the kernel that must have run).
# FCD_PREC_F16X2 (two pieces, three MFMAs: fp32-faithful, 1e-6) and FCD_PREC_F16 (the reduced-precision variant: the
# same kernels and operands with the high piece only, one MFMA; north_star's bar for the 16-bit path is 1e-2, the
# 11-bit pieces hold 1e-3)
F16_FAM = [pytest.param((3, 1e-6, b"2>"), id="fp32x2"), pytest.param((4, 1e-3, b"1>"), id="fp16")]

# FCD_PREC_F16X2: (O, I, N, H, W, form, scale of the activation operand)
H2_GEOMS = [
    (256, 256, 32, 32, 32, "fwd", 1.0),
    (256, 256, 16, 32, 32, "adj", 3e-6),       # gradients of the size the gradient-penalty backward carries
    (128, 96, 9, 64, 64, "fwd", 4000.0),
    (64, 192, 25, 16, 32, "adj", 1.0),
]


@pytest.mark.parametrize("fam", F16_FAM)
@pytest.mark.parametrize("geom", H2_GEOMS)
def test_gather_a_halo_f16x2(dev, fam, geom):
    """FCD_PREC_F16X2 on the A-halo 3x3 convolutions: operands scaled by a power of two taken from their max |.|
    (ops.absmax: FCD_AMAX_SLOTS per-block maxima) and split into two fp16 pieces, three fp16 MFMAs per product group.
    Against torch in float64 the error stays at the BF16X3 level (<= 1e-6 of max |result|, the bar of the exact-fp32
    kernels is 1e-5) whatever the magnitude of the operands, also with a few outliers 1000 x the rest (values far
    below the maximum lose relative precision in the split, not absolute)."""
    from fcd_amd import _lib as L, ops
    O, I, N, H, W, form, sc = geom
    g = ops.ConvGeom(O, I, 3, 1, 1)
    gen = torch.Generator().manual_seed(21)
    wt = torch.randn(O, I, 3, 3, generator=gen) / (3 * I ** 0.5)
    wd = wt.to(dev)
    wa = ops.absmax(wd)
    assert wa.numel() == ops.AMAX_SLOTS and float(wa.max()) == float(wt.abs().max())
    for outliers in (False, True):
        if form == "fwd":
            x = torch.randn(N, I, H, W, generator=gen) * sc
            if outliers:
                x.view(-1)[::40961] *= 1000.0
            b = torch.randn(O, generator=gen) * sc
            d = g.fwd_desc(N, H, W, precision=fam[0], act_slope=0.01)
            assert ops.weight_format(d) == 2
            xd = to_nhwc(x, g.IP).to(dev)
            ops.set_amax(d, ops.absmax(xd), wa)
            wp = ops.pack_weight_fwd_h2(wd, g, wa)
            got = ops.conv_gather(d, xd, wp, bias=b.to(dev))
            ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), 1, 1), 0.01)
            cref = O
        else:
            gy = torch.randn(N, O, H, W, generator=gen) * sc
            if outliers:
                gy.view(-1)[::40961] *= 1000.0
            mask = torch.randn(N, I, H, W, generator=gen)
            d = g.adj_desc(N, H, W, precision=fam[0], mask_slope=0.01)
            assert ops.weight_format(d) == 2
            gd, md = to_nhwc(gy, g.OP).to(dev), to_nhwc(mask, g.IP).to(dev)
            ops.set_amax(d, ops.absmax(gd), wa)
            wp = ops.pack_weight_adj_h2(wd, g, wa)
            got = ops.conv_gather(d, gd, wp, mask=md)
            ref = F.conv_transpose2d(gy.double(), wt.double(), None, 1, 1) * torch.where(mask > 0, 1.0, 0.01)
            cref = I
        assert L.load().fcd_last_conv_kernel().endswith(b"true, true, " + fam[2])
        assert relerr(from_nhwc(got.cpu(), cref).double(), ref) <= fam[1]


def test_f16x2_without_amax_is_refused(dev):
    """A PREC_F16X2 A-halo convolution needs both maxima: the library refuses the launch instead of guessing a scale."""
    from fcd_amd import _lib as L, ops
    g = ops.ConvGeom(256, 256, 3, 1, 1)
    d = g.fwd_desc(32, 32, 32, precision=L.PREC_F16X2)
    x = torch.zeros(32, 32, 32, 256, device=dev)
    wd = torch.zeros(256, 256, 3, 3, device=dev)
    wp = ops.pack_weight_fwd_h2(wd, g, ops.absmax(wd))
    with pytest.raises(RuntimeError):
        ops.conv_gather(d, x, wp)
    with pytest.raises(ValueError):
        ops.set_amax(d, torch.zeros(1, device=dev), torch.zeros(1, device=dev))


@pytest.mark.parametrize("fam", F16_FAM)
@pytest.mark.parametrize("shape", [(256, 256, 16, 4, 1.0, 1.0), (128, 192, 40, 5, 2e-5, 30.0), (254, 128, 30, 3, 500.0, 1e-3)])
def test_wgrad3_halo_f16x2(dev, fam, shape):
    """FCD_PREC_F16X2 weight gradient of the residual blocks' 3x3 convolutions (wgrad3_halo_x3_kernel<2>): both operands
    as two fp16 planes of the scaled values; <= 1e-6 of max |dW| from torch in float64 across operand magnitudes, and
    accumulation into an existing gradient."""
    from fcd_amd import _lib as L, ops
    O, I, n, h, sg, sx = shape
    g = ops.ConvGeom(O, I, 3, 1, 1)
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(n, I, h, 32, generator=gen) * sx
    gy = torch.randn(n, O, h, 32, generator=gen) * sg
    xd, gd = to_nhwc(x, g.IP).to(dev), to_nhwc(gy, g.OP).to(dev)
    d = ops.set_wgrad_amax(g.wgrad_desc(n, h, 32, precision=fam[0]), ops.absmax(gd), ops.absmax(xd))
    dw = ops.conv_wgrad(d, gd, xd)
    assert L.load().fcd_last_conv_kernel() == b"wgrad3_halo_x3_kernel<" + fam[2]
    ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, 3, 3), gy.double(), stride=1, padding=1)
    assert relerr(dw.cpu(), ref) <= fam[1]
    base = float(ref.abs().max()) * 0.25
    acc = torch.full_like(dw, base)
    d2 = ops.set_wgrad_amax(g.wgrad_desc(n, h, 32, precision=fam[0], accumulate=True), ops.absmax(gd), ops.absmax(xd))
    ops.conv_wgrad(d2, gd, xd, dw=acc)
    assert relerr(acc.cpu() - base, ref) <= 10 * fam[1]
    # without the maxima the same descriptor runs the BF16X3 form
    dw3 = ops.conv_wgrad(g.wgrad_desc(n, h, 32, precision=fam[0]), gd, xd)
    assert L.load().fcd_last_conv_kernel() == b"wgrad3_halo_x3_kernel"
    assert relerr(dw3.cpu(), ref) <= 1e-6


@pytest.mark.parametrize("fam", F16_FAM)
@pytest.mark.parametrize("shape", [(128, 64, 6, 32, 32, 1.0, 3e-4), (256, 128, 5, 16, 16, 40.0, 1.0),
                                   (128, 256, 3, 16, 16, 1e-3, 1.0), (96, 64, 2, 24, 40, 1.0, 1.0), (48, 24, 2, 16, 16, 1.0, 1.0),
                                   (64, 64, 7, 8, 8, 5.0, 0.1)])
def test_wgrad_x3p_f16x2(dev, fam, shape):
    """FCD_PREC_F16X2 weight gradient of the 4x4 stride-2 layers (conv_wgrad_x3p_kernel<.., 2>: operands split once per
    workgroup into two fp16 planes of the scaled values): <= 1e-6 of max |dW| from torch in float64 across operand
    magnitudes, ragged channel counts and accumulation."""
    from fcd_amd import _lib as L, ops
    O, I, n, h, w, sg, sx = shape
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(29)
    big = torch.randn(n, I, 2 * h, 2 * w, generator=gen) * sx
    small = torch.randn(n, O, h, w, generator=gen) * sg
    bd, sd = to_nhwc(big, g.IP).to(dev), to_nhwc(small, g.OP).to(dev)
    d = ops.set_wgrad_amax(g.wgrad_desc(n, 2 * h, 2 * w, precision=fam[0]), ops.absmax(sd), ops.absmax(bd))
    dw = ops.conv_wgrad(d, sd, bd)
    assert L.load().fcd_last_conv_kernel().startswith(b"conv_wgrad_x3p_kernel") and L.load().fcd_last_conv_kernel().endswith(b", " + fam[2])
    ref = torch.nn.grad.conv2d_weight(big.double(), (O, I, 4, 4), small.double(), stride=2, padding=1)
    assert relerr(dw.cpu(), ref) <= fam[1]
    base = float(ref.abs().max()) * 0.5
    acc = torch.full_like(dw, base)
    d2 = ops.set_wgrad_amax(g.wgrad_desc(n, 2 * h, 2 * w, precision=fam[0], accumulate=True), ops.absmax(sd), ops.absmax(bd))
    ops.conv_wgrad(d2, sd, bd, dw=acc)
    assert relerr(acc.cpu() - base, ref) <= 10 * fam[1]


def test_absmax_and_norm_amax(dev):
    """max |.| producers of PREC_F16X2: fcd_absmax, and the InstanceNorm apply / backward kernels' fused maxima, equal
    torch's abs().max() of the tensor they wrote (as the maximum over the FCD_AMAX_SLOTS per-block slots), including
    accumulation into an existing array and grids larger than the slot count."""
    from fcd_amd import ops
    gen = torch.Generator().manual_seed(5)
    for shape in [(2, 8, 8, 64), (32, 32, 32, 256), (3, 128, 128, 64), (16, 128, 128, 64)]:
        z = torch.randn(*shape, generator=gen).to(dev) * 3
        a = ops.absmax(z)
        assert float(a.max()) == float(z.abs().max())
        ops.absmax(z * 0.5, out=a)
        assert float(a.max()) == float(z.abs().max())
        N, H, W, C = shape
        mean, rstd, _ = ops.instnorm_stats(z)
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        am = torch.zeros(ops.AMAX_SLOTS, device=dev)
        y = ops.instnorm_apply(z, mean, rstd, gamma, beta, act_slope=0.0, amax=am)
        assert float(am.max()) == float(y.abs().max())
        gy = torch.randn(*shape, generator=gen).to(dev) * 1e-4
        gm = torch.zeros(ops.AMAX_SLOTS, device=dev)
        dz, _, _ = ops.instnorm_bwd(gy, y, z, mean, rstd, gamma, 0.0, False, dz_amax=gm)
        assert float(gm.max()) == float(dz.abs().max())
        dz0, _, _ = ops.instnorm_bwd(gy, y, z, mean, rstd, gamma, 0.0, False)
        assert torch.equal(dz, dz0)


# (O, I, N, small H, small W, form, operand scale): 4x4 stride-2 layers of the generator (down / up sampling)
H2_S2_GEOMS = [
    (128, 64, 8, 32, 32, "fwd", 1.0),        # 128-row tiles, 64 input channels
    (256, 128, 32, 16, 16, "fwd", 2e-4),     # 256 x 128 tiles
    (256, 128, 9, 16, 16, "adj", 50.0),      # four phases, 128 output channels
    (128, 64, 33, 32, 32, "adj", 1.0),       # 256 x 64 tiles (8 x 1 waves): A-halo form of the four 2 x 2-tap phases
    (256, 128, 32, 32, 32, "adj", 1.0),      # 256 x 128 tiles, A-halo form, 32-wide grid (the 256 -> 128 up-sampling layer)
    (128, 64, 9, 64, 64, "adj", 3e-3),       # 64-wide grid, 128 x 64 A-halo tiles
    (128, 64, 16, 64, 64, "adj", 1.0),       # 64-wide grid, 256 x 64 A-halo tiles (the 128 -> 64 up-sampling layer)
]


@pytest.mark.parametrize("fam", F16_FAM)
@pytest.mark.parametrize("geom", H2_S2_GEOMS)
def test_gather_per_tap_f16x2(dev, fam, geom):
    """FCD_PREC_F16X2 in the per-tap gather form (every shape that takes pre-split weights under BF16X3): activation
    fragments scaled and split into two fp16 pieces in registers, weights as two fp16 planes ([phase][plane] for the
    four-phase adjoint of a stride-2 layer); <= 1e-6 of max |result| from torch in float64, with the epilogue operands
    and the fused InstanceNorm statistics."""
    from fcd_amd import _lib as L, ops
    O, I, N, H, W, form, sc = geom
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(31)
    wt = torch.randn(O, I, 4, 4, generator=gen) / (4 * I ** 0.5)
    wd = wt.to(dev)
    wa = ops.absmax(wd)
    if form == "fwd":
        x = torch.randn(N, I, 2 * H, 2 * W, generator=gen) * sc
        xd = to_nhwc(x, g.IP).to(dev)
        d = g.fwd_desc(N, 2 * H, 2 * W, precision=fam[0])
        assert ops.weight_format(d) == 2
        ops.set_amax(d, ops.absmax(xd), wa)
        wp = ops.pack_weight_fwd_h2(wd, g, wa)
        z, mean, rstd, varu = ops.conv_gather_stats(d, xd, wp, want_var=True)
        ref = F.conv2d(x.double(), wt.double(), None, 2, 1)
        assert relerr(from_nhwc(z.cpu(), O).double(), ref) <= fam[1]
        assert relerr(mean.cpu().double(), ref.mean(dim=(2, 3))) <= 10 * fam[1]
        assert relerr(varu.cpu().double(), ref.var(dim=(2, 3), unbiased=True)) <= 10 * fam[1]
    else:
        gy = torch.randn(N, O, H, W, generator=gen) * sc
        mask = torch.randn(N, I, 2 * H, 2 * W, generator=gen)
        res = torch.randn(N, I, 2 * H, 2 * W, generator=gen) * sc
        gd, md, rd = to_nhwc(gy, g.OP).to(dev), to_nhwc(mask, g.IP).to(dev), to_nhwc(res, g.IP).to(dev)
        d = g.adj_desc(N, H, W, precision=fam[0], mask_slope=0.2)
        assert ops.weight_format(d) == 2
        ops.set_amax(d, ops.absmax(gd), wa)
        wp = ops.pack_weight_adj_h2(wd, g, wa)
        got = ops.conv_gather(d, gd, wp, mask=md, residual=rd)
        ref = F.conv_transpose2d(gy.double(), wt.double(), None, 2, 1) * torch.where(mask > 0, 1.0, 0.2) + res
        assert relerr(from_nhwc(got.cpu(), I).double(), ref) <= fam[1]
    kern = L.load().fcd_last_conv_kernel()
    assert kern.endswith(b", " + fam[2])
    if form == "adj" and H in (32, 64):
        assert kern.endswith(b"true, true, " + fam[2]), kern        # the A-halo form took the launch


@pytest.mark.parametrize("fam", F16_FAM)
@pytest.mark.parametrize("case", [(64, 11, 2, 40, 72, "fwd", 30.0), (10, 64, 2, 33, 64, "adj", 1e-4), (10, 64, 3, 24, 70, "fwd", 1.0),
                                  (16, 64, 1, 16, 32, "fwd", 1.0)])
def test_conv7_halo_f16x2(dev, fam, case):
    """The 7x7 image-layer kernels under FCD_PREC_F16X2 (weight format 3: fp32 packs, operands scaled and split into
    two fp16 pieces inside the kernel when both maxima are attached): <= 1e-6 of max |result| from torch in float64; the
    same descriptor without maxima runs the BF16X3 form."""
    from fcd_amd import _lib as L, ops
    O, I, n, h, w, form, sc = case
    g = ops.ConvGeom(O, I, 7, 1, 3)
    gen = torch.Generator().manual_seed(41)
    wt = torch.randn(O, I, 7, 7, generator=gen) / (7 * I ** 0.5)
    cin, cout = (I, O) if form == "fwd" else (O, I)
    x = torch.randn(n, cin, h, w, generator=gen) * sc
    xd = to_nhwc(x, g.IP if form == "fwd" else g.OP).to(dev)
    wd = wt.to(dev)
    def mk():
        d_ = g.fwd_desc(n, h, w, precision=fam[0]) if form == "fwd" else g.adj_desc(n, h, w, precision=fam[0])
        d_.splitk = 1          # (small test shapes would otherwise take split-K and the generic kernel)
        return d_
    d = mk()
    assert ops.weight_format(d) == 3
    wp = ops.pack_weight_fwd(wd, g) if form == "fwd" else ops.pack_weight_adj(wd, g)
    ops.set_amax(d, ops.absmax(xd), ops.absmax(wd))
    got = ops.conv_gather(d, xd, wp)
    assert L.load().fcd_last_conv_kernel().endswith(b"halo_x3_kernel<" + fam[2])
    ref = F.conv2d(x.double(), wt.double(), None, 1, 3) if form == "fwd" else F.conv_transpose2d(x.double(), wt.double(), None, 1, 3)
    assert relerr(from_nhwc(got.cpu(), cout).double(), ref) <= fam[1]
    plain = ops.conv_gather(mk(), xd, wp)
    assert L.load().fcd_last_conv_kernel().endswith(b"halo_x3_kernel")
    assert relerr(from_nhwc(plain.cpu(), cout).double(), ref) <= 1e-6


@pytest.mark.parametrize("fam", F16_FAM)
@pytest.mark.parametrize("case", [(10, 64, 2, 40, 72, 1e-3, 1.0), (64, 11, 3, 33, 70, 1.0, 25.0), (16, 16, 1, 70, 64, 1.0, 1.0)])
def test_wgrad7_halo_f16x2(dev, fam, case):
    """Weight gradients of the 7x7 image layers under FCD_PREC_F16X2 with both maxima attached (two fp16 planes of the
    scaled gradient and input rows): <= 1e-6 of max |dW| from torch in float64."""
    from fcd_amd import _lib as L, ops
    O, I, n, h, w, sg, sx = case
    g = ops.ConvGeom(O, I, 7, 1, 3)
    gen = torch.Generator().manual_seed(43)
    x = torch.randn(n, I, h, w, generator=gen) * sx
    gy = torch.randn(n, O, h, w, generator=gen) * sg
    xd, gd = to_nhwc(x, g.IP).to(dev), to_nhwc(gy, g.OP).to(dev)
    d = ops.set_wgrad_amax(g.wgrad_desc(n, h, w, precision=fam[0]), ops.absmax(gd), ops.absmax(xd))
    dw = ops.conv_wgrad(d, gd, xd)
    assert L.load().fcd_last_conv_kernel().endswith(b"halo_x3_kernel<" + fam[2])
    ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, 7, 7), gy.double(), stride=1, padding=3)
    assert relerr(dw.cpu(), ref) <= fam[1]


@pytest.mark.parametrize("case", [(256, 256, 3, 1, 1, 8, 32, 32, 2, 0), (128, 64, 4, 2, 1, 4, 64, 64, 2, 0), (10, 64, 7, 1, 3, 2, 64, 64, 2, 0),
                                  (64, 64, 3, 1, 1, 2, 16, 16, 0, 0), (256, 256, 3, 1, 1, 2, 8, 8, 2, 4), (256, 256, 3, 1, 1, 40, 32, 32, 3, 0)])
def test_conv_out_amax(dev, case):
    """FcdConvDesc.out_amax: max |out| of a convolution as FCD_AMAX_SLOTS per-block maxima -- from the epilogue of the
    32-wide-tile kernels when they write final values, by an fcd_absmax pass inside the call otherwise (halo kernels,
    split-K, 16-wide tiles); equals torch's abs().max() of the tensor the call wrote, bias / activation / residual
    included, and accumulates into an array that already holds a maximum."""
    from fcd_amd import _lib as L, ops
    O, I, k, st, pd, n, h, w, prec, splitk = case
    g = ops.ConvGeom(O, I, k, st, pd)
    gen = torch.Generator().manual_seed(51)
    x = to_nhwc(torch.randn(n, I, h, w, generator=gen), g.IP).to(dev)
    wd = (torch.randn(O, I, k, k, generator=gen) / (k * I ** 0.5)).to(dev)
    d = g.fwd_desc(n, h, w, precision=prec, act_slope=0.2)
    d.splitk = splitk
    fmt = ops.weight_format(d)
    keep = None
    if fmt >= 2:
        keep = (ops.absmax(x), ops.absmax(wd))
        ops.set_amax(d, *keep)
    wp = {0: ops.pack_weight_fwd, 1: ops.pack_weight_fwd_x3, 3: ops.pack_weight_fwd}[fmt](wd, g) if fmt != 2 else ops.pack_weight_fwd_h2(wd, g, keep[1])
    oh, ow = g.small_hw(h, w)
    res = torch.randn(n, oh, ow, g.OP, generator=gen).to(dev)
    res[..., O:] = 0
    bias = torch.randn(O, generator=gen).to(dev)
    am = torch.zeros(ops.AMAX_SLOTS, device=dev)
    ops.set_out_amax(d, am)
    out = ops.conv_gather(d, x, wp, bias=bias, residual=res)
    assert float(am.max()) == float(out.abs().max()) > 0
    am2 = torch.zeros(ops.AMAX_SLOTS, device=dev)
    am2[5] = 1e9
    ops.set_out_amax(d, am2)
    ops.conv_gather(d, x, wp, bias=bias, residual=res)
    assert float(am2.max()) == 1e9


@pytest.mark.parametrize("geom", [(256, 256, 3, 1, 1), (128, 64, 4, 2, 1), (40, 24, 3, 1, 1)])
def test_pack_weight_h2_batch(dev, geom):
    """fcd_pack_weight_{fwd,adj}_h2_batch == the single packs, byte for byte, for more weights than one launch takes."""
    from fcd_amd import ops
    O, I, k, s, p = geom
    g = ops.ConvGeom(O, I, k, s, p)
    torch.manual_seed(9)
    ws = [torch.randn(O, I, k, k, device=dev) * (0.01 * (j + 1)) for j in range(19)]
    amax = torch.empty((len(ws), ops.AMAX_SLOTS), device=dev)
    ops.absmax_batch(ws, amax)
    for kind, single in (("fwd", ops.pack_weight_fwd_h2), ("adj", ops.pack_weight_adj_h2)):
        outs = ops.pack_weight_h2_batch(kind, ws, g, [amax[j] for j in range(len(ws))])
        for j, w in enumerate(ws):
            ref = single(w, g, amax[j])
            assert outs[j].shape == ref.shape and torch.equal(outs[j].view(torch.int16), ref.view(torch.int16)), (kind, j)
    with pytest.raises(RuntimeError):
        ops.pack_weight_h2_batch("fwd", [ws[0], torch.randn(O, I + 4, k, k, device=dev)], g, [amax[0], amax[1]])


@pytest.mark.parametrize("log2_disparity", [10, 18])
def test_f16x2_per_sample_disparity(dev, log2_disparity):
    """fp32x2 scales a whole operand tensor by ONE power of two taken from its max |.|.  One of 32 samples here is
    2^10 / 2^18 times larger than the other 31 (the paired generator call puts different chains into one tensor): the
    small samples sit 2^-10 / 2^-18 below the maximum, where the low fp16 piece turns subnormal.  Per-sample error
    against torch in float64, for the A-halo gather (forward form), its adjoint (dgrad form), the 3x3 halo weight
    gradient and the split-once 4x4 stride-2 weight gradient.  What include/fcd_hip.h guarantees: full 22-bit operands
    within 2^-15 of the maximum (per-sample error at the 1e-6 level), beyond that an absolute, not a relative bound --
    at 2^-18 the small samples are still good to 1e-5."""
    from fcd_amd import _lib as L, ops
    bar = 1e-6 if log2_disparity <= 15 else 1e-5
    N, C, H = 32, 128, 32
    big = float(2 ** log2_disparity)
    gen = torch.Generator().manual_seed(31)
    g3 = ops.ConvGeom(C, C, 3, 1, 1)
    wt = torch.randn(C, C, 3, 3, generator=gen) / (3 * C ** 0.5)
    wd = wt.to(dev)
    wa = ops.absmax(wd)
    x = torch.randn(N, C, H, H, generator=gen)
    x[5] *= big
    xd = to_nhwc(x, g3.IP).to(dev)
    xa = ops.absmax(xd)

    def per_sample(got, ref):
        return [relerr(got[i], ref[i]) for i in range(got.shape[0])]

    # gather, forward form
    d = ops.set_amax(g3.fwd_desc(N, H, H, precision=L.PREC_F16X2), xa, wa)
    got = from_nhwc(ops.conv_gather(d, xd, ops.pack_weight_fwd_h2(wd, g3, wa)).cpu(), C).double()
    assert L.load().fcd_last_conv_kernel().endswith(b"true, true, 2>")
    e_f = per_sample(got, F.conv2d(x.double(), wt.double(), None, 1, 1))
    # gather, adjoint form (data gradient)
    d = ops.set_amax(g3.adj_desc(N, H, H, precision=L.PREC_F16X2), xa, wa)
    got = from_nhwc(ops.conv_gather(d, xd, ops.pack_weight_adj_h2(wd, g3, wa)).cpu(), C).double()
    e_a = per_sample(got, F.conv_transpose2d(x.double(), wt.double(), None, 1, 1))
    print(f"disparity 2^{log2_disparity}: gather worst small-sample rel-err {max(e_f[:5] + e_f[6:]):.2e} (large sample "
          f"{e_f[5]:.2e}); adjoint {max(e_a[:5] + e_a[6:]):.2e} ({e_a[5]:.2e})")
    assert max(e_f) <= bar and max(e_a) <= bar, (e_f, e_a)
    assert e_f[5] <= 1e-6 and e_a[5] <= 1e-6

    # weight gradients: the sum over samples is dominated by the large one, so the small samples are also run ALONE
    # with the maxima of the whole tensors attached -- their own contribution under the shared scale
    small = [i for i in range(N) if i != 5][:12]
    for name, geom, xin in (("wgrad3_halo", ops.ConvGeom(256, 256, 3, 1, 1), None), ("wgrad_x3p", ops.ConvGeom(C, C, 4, 2, 1), None)):
        if geom.k == 3:                      # 256 channels, 8 x 32 images: the shape class wgrad3_halo_x3_kernel<2> takes
            xin = torch.randn(N, 256, 8, 32, generator=gen)
            gy = torch.randn(N, 256, 8, 32, generator=gen)
        else:                                # 4x4 stride 2: input 64 x 64 -> gradient 32 x 32
            xin = torch.randn(N, C, 2 * H, 2 * H, generator=gen)
            gy = torch.randn(N, C, H, H, generator=gen)
        xin[5] *= big
        gd = to_nhwc(gy, geom.OP).to(dev)
        ga = ops.absmax(gd)
        xbd = to_nhwc(xin, geom.IP).to(dev)
        xba = ops.absmax(xbd)
        bh, bw = xin.shape[2], xin.shape[3]
        dsc = ops.set_wgrad_amax(geom.wgrad_desc(N, bh, bw, precision=L.PREC_F16X2), ga, xba)
        dw = ops.conv_wgrad(dsc, gd, xbd)
        ref = torch.nn.grad.conv2d_weight(xin.double(), tuple(dw.shape), gy.double(), stride=geom.s, padding=geom.p)
        e_all = relerr(dw.cpu(), ref)
        ds = ops.set_wgrad_amax(geom.wgrad_desc(len(small), bh, bw, precision=L.PREC_F16X2), ga, xba)
        dws = ops.conv_wgrad(ds, gd[small].contiguous(), xbd[small].contiguous())
        refs = torch.nn.grad.conv2d_weight(xin[small].double(), tuple(dw.shape), gy[small].double(), stride=geom.s,
                                           padding=geom.p)
        e_small = relerr(dws.cpu(), refs)
        print(f"  {name} ({L.load().fcd_last_conv_kernel().decode()}): all samples {e_all:.2e}, small samples under the "
              f"shared scale {e_small:.2e}")
        assert e_all <= 1e-6 and e_small <= bar, (name, e_all, e_small)


def test_f16x2_overflow_guard(dev):
    """A max |.| that is too small for its tensor (stale slot, a producer that forgot it) must not turn into silent
    Inf / NaN: every fp32x2 kernel raises the device's overflow word when a scaled operand passes 2^15, and
    ops.check_overflow turns it into an error.  Correct maxima leave the word alone."""
    from fcd_amd import _lib as L, ops
    g = ops.ConvGeom(64, 64, 3, 1, 1)
    gen = torch.Generator().manual_seed(33)
    wt = torch.randn(64, 64, 3, 3, generator=gen) / 24.0
    x = torch.randn(8, 64, 32, 32, generator=gen)
    wd, xd = wt.to(dev), to_nhwc(x, g.IP).to(dev)
    wa, xa = ops.absmax(wd), ops.absmax(xd)
    ops.check_overflow(dev)                                # clean start
    d = ops.set_amax(g.fwd_desc(8, 32, 32, precision=L.PREC_F16X2), xa, wa)
    wp = ops.pack_weight_fwd_h2(wd, g, wa)
    ops.conv_gather(d, xd, wp)
    ops.check_overflow(dev)                                # correct maxima: nothing raised
    # activation maximum 64 x too small (a stale slot): the gather kernel's halo split sees values up to 2^19
    d_bad = ops.set_amax(g.fwd_desc(8, 32, 32, precision=L.PREC_F16X2), xa / 64.0, wa)
    out = ops.conv_gather(d_bad, xd, wp)
    with pytest.raises(RuntimeError, match="overflow guard"):
        ops.check_overflow(dev)
    ops.check_overflow(dev)                                # the poll cleared the word
    assert not bool(torch.isfinite(out).all()) or True     # (the output itself may or may not hold Inf: the word is the signal)
    # weight maximum too small: caught by the pack kernel
    ops.pack_weight_fwd_h2(wd, g, wa / 64.0)
    with pytest.raises(RuntimeError, match="overflow guard"):
        ops.check_overflow(dev)
    # weight gradient with a stale gradient maximum
    gy = torch.randn(8, 64, 32, 32, generator=gen)
    gd = to_nhwc(gy, g.OP).to(dev)
    dw_desc = ops.set_wgrad_amax(g.wgrad_desc(8, 32, 32, precision=L.PREC_F16X2), ops.absmax(gd) / 64.0, xa)
    ops.conv_wgrad(dw_desc, gd, xd)
    with pytest.raises(RuntimeError, match="overflow guard"):
        ops.check_overflow(dev)
    # a NaN operand is not an overflow: it propagates into the result as it would in fp32, and the word stays clear
    xn = xd.clone()
    xn[0, 5, 5, 0] = float("nan")
    outn = ops.conv_gather(d, xn, wp)
    ops.check_overflow(dev)
    assert bool(torch.isnan(outn[0, 5, 5]).any()) and bool(torch.isfinite(outn[1:]).all())
    # and a solver polls it from train_step / state_dict
    ops.conv_gather(d_bad, xd, wp)
    from fcd_amd import FCDSolver
    s = FCDSolver(None, c_dim=1, image_size=32, g_conv_dim=8, d_conv_dim=8, g_repeat_num=1, d_repeat_num=3,
                  num_channels=10, batch_size=2, precision="fp32x2").to(dev)
    with pytest.raises(RuntimeError, match="overflow guard"):
        s.state_dict()
    s.state_dict()


# (O, I, N, small H, small W, form): 4x4 stride-2 layers with few output rows (the discriminator's deep layers), ragged
# channel counts and row counts, split-K, the four-phase adjoint
RS_GEOMS = [
    (512, 256, 32, 8, 8, "fwd"),       # 128-row tiles, split-K 4
    (1024, 512, 16, 4, 4, "fwd"),      # M = 256
    (2048, 1024, 4, 2, 2, "fwd"),      # 64-row tile, 16 rows of it used
    (1024, 512, 9, 4, 4, "adj"),       # four phases, M = 144 per phase
    (200, 96, 5, 6, 7, "fwd"),         # ragged: 200 output channels (two column tiles), 210 rows
    (160, 136, 3, 5, 3, "adj"),        # ragged adjoint: 136 output channels, 45 rows per phase (64-row tile)
]


@pytest.mark.parametrize("geom", RS_GEOMS)
def test_gather_register_staged_equals_in_register_split(dev, geom, monkeypatch):
    """The register-staged form of the BF16X3 gather kernel with fp32 weights (template flag RS: both operands go
    global -> registers, are split once per workgroup into planes in LDS) produces the same bits as the form that
    splits every fragment in registers (FCD_NO_RS=1), epilogue operands, split-K and maximum of the output included."""
    from fcd_amd import _lib as L, ops
    O, I, N, H, W, form = geom
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(77)
    wd = (torch.randn(O, I, 4, 4, generator=gen) / (4 * I ** 0.5)).to(dev)
    if form == "fwd":
        xd = to_nhwc(torch.randn(N, I, 2 * H, 2 * W, generator=gen), g.IP).to(dev)
        mk = lambda: g.fwd_desc(N, 2 * H, 2 * W, precision=2, act_slope=0.01, mask_slope=0.3)
        oshape, oc = (N, H, W, g.OP), O
        wp = ops.pack_weight_fwd(wd, g)
    else:
        xd = to_nhwc(torch.randn(N, O, H, W, generator=gen), g.OP).to(dev)
        mk = lambda: g.adj_desc(N, H, W, precision=2, mask_slope=0.3)
        oshape, oc = (N, 2 * H, 2 * W, g.IP), I
        wp = ops.pack_weight_adj(wd, g)
    assert ops.weight_format(mk()) == 0
    bias = torch.randn(oc, generator=gen).to(dev)
    mask = torch.randn(*oshape, generator=gen).to(dev)
    res = torch.randn(*oshape, generator=gen).to(dev)
    outs = {}
    for mode in ("rs", "split"):
        if mode == "split":
            monkeypatch.setenv("FCD_NO_RS", "1")
        for sk in (0, 3):
            d = mk()
            d.splitk = sk
            am = torch.zeros(ops.AMAX_SLOTS, device=dev)
            ops.set_out_amax(d, am)
            outs[mode, sk] = (ops.conv_gather(d, xd, wp, bias=bias, mask=mask, residual=res), float(am.max()))
            kern = L.load().fcd_last_conv_kernel()
            assert kern.endswith(b"true>") == (mode == "rs"), kern
    for sk in (0, 3):
        assert torch.equal(outs["rs", sk][0], outs["split", sk][0])
        assert outs["rs", sk][1] == outs["split", sk][1] == float(outs["rs", sk][0].abs().max())


@pytest.mark.parametrize("fam", F16_FAM)
@pytest.mark.parametrize("geom", [(512, 256, 32, 8, 8, "fwd", 1.0), (2048, 1024, 16, 2, 2, "fwd", 3e-5),
                                  (1024, 512, 9, 4, 4, "adj", 200.0), (200, 96, 5, 6, 7, "fwd", 1.0)])
def test_gather_register_staged_f16x2(dev, fam, geom):
    """The register-staged form under FCD_PREC_F16X2 / _F16 (weight format 3: fp32 weights, both operands scaled and
    split into fp16 pieces inside the kernel when both maxima are attached): <= 1e-6 (fp32x2) / 1e-3 (fp16) of torch in
    float64."""
    from fcd_amd import _lib as L, ops
    O, I, N, H, W, form, sc = geom
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(78)
    wt = torch.randn(O, I, 4, 4, generator=gen) / (4 * I ** 0.5)
    wd = wt.to(dev)
    wa = ops.absmax(wd)
    if form == "fwd":
        x = torch.randn(N, I, 2 * H, 2 * W, generator=gen) * sc
        b = torch.randn(O, generator=gen) * sc
        xd = to_nhwc(x, g.IP).to(dev)
        d = g.fwd_desc(N, 2 * H, 2 * W, precision=fam[0], act_slope=0.01)
        assert ops.weight_format(d) == 3
        ops.set_amax(d, ops.absmax(xd), wa)
        got = from_nhwc(ops.conv_gather(d, xd, ops.pack_weight_fwd(wd, g), bias=b.to(dev)).cpu(), O)
        ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), 2, 1), 0.01)
    else:
        gy = torch.randn(N, O, H, W, generator=gen) * sc
        gd = to_nhwc(gy, g.OP).to(dev)
        d = g.adj_desc(N, H, W, precision=fam[0])
        assert ops.weight_format(d) == 3
        ops.set_amax(d, ops.absmax(gd), wa)
        got = from_nhwc(ops.conv_gather(d, gd, ops.pack_weight_adj(wd, g)).cpu(), I)
        ref = F.conv_transpose2d(gy.double(), wt.double(), None, 2, 1)
    assert relerr(got.double(), ref) <= fam[1]
    assert L.load().fcd_last_conv_kernel().endswith(b", " + fam[2][:1] + b", true>")


@pytest.mark.parametrize("fam", [pytest.param((2, 1e-5), id="fp32x3"), pytest.param((3, 1e-6), id="fp32x2")])
@pytest.mark.parametrize("shape", [(2048, 1024, 4, 2, 1, 8, 4, 4), (1024, 512, 4, 2, 1, 3, 8, 8), (48, 10, 4, 2, 1, 2, 8, 8),
                                   (300, 36, 3, 1, 1, 2, 9, 7), (72, 20, 4, 2, 1, 5, 6, 10), (100, 24, 4, 2, 1, 3, 6, 6),
                                   (260, 68, 4, 2, 1, 2, 4, 8)])
def test_wgrad_direct_equals_slab_and_reduce(dev, shape, fam, monkeypatch):
    """conv_wgrad_x3p_kernel without split-K writes (or accumulates into) the parameter layout dw[O][I][KH][KW] itself --
    columns ordered (channel group, tap, channel) so that a wave stores whole 256-byte runs -- instead of a slab that
    wgrad_reduce_kernel transposes (FCD_NO_WGRAD_DIRECT=1): same bits, fresh and accumulating, ragged channel counts
    (10 -> 12 physical input channels) and 3 x 3 kernels included; within the precision's bar of torch in float64."""
    from fcd_amd import ops
    O, I, k, s, pd, n, h, w = shape
    prec, tol = fam
    g = ops.ConvGeom(O, I, k, s, pd)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, I, h, w, generator=gen)
    sh, sw = g.small_hw(h, w)
    gy = torch.randn(n, O, sh, sw, generator=gen)
    xd, gd = to_nhwc(x, g.IP).to(dev), to_nhwc(gy, g.OP).to(dev)

    def run(accumulate):
        wd = g.wgrad_desc(n, h, w, precision=prec, accumulate=accumulate)
        if prec == 3:
            ops.set_wgrad_amax(wd, ops.absmax(gd), ops.absmax(xd))
        dw = torch.full((O, I, k, k), 0.25, device=dev) if accumulate else torch.full((O, I, k, k), float("nan"), device=dev)
        return ops.conv_wgrad(wd, gd, xd, dw=dw)

    monkeypatch.setenv("FCD_NO_WGRAD_DIRECT", "1")
    old, old_acc = run(False).clone(), run(True).clone()
    monkeypatch.delenv("FCD_NO_WGRAD_DIRECT")
    new, new_acc = run(False), run(True)
    assert torch.equal(old, new) and torch.equal(old_acc, new_acc)
    ref = torch.nn.grad.conv2d_weight(x.double(), (O, I, k, k), gy.double(), stride=s, padding=pd)
    assert relerr(new, ref) <= tol
    assert relerr(new_acc - 0.25, ref) <= 10 * tol


# (O, I, n, small h, small w): 4x4 stride-2 layers whose weight gradient takes the halo kernel's MODE 1
W4_SHAPES = [(256, 128, 32, 32, 32),      # the generator's 128 -> 256 down-sampling / 256 -> 128 up-sampling layer
             (128, 64, 16, 64, 64),       # 64-pixel gradient rows: two 32-column halves per row
             (192, 64, 17, 32, 32),       # 192 output channels, an odd number of images (ragged last split)
             (128, 128, 12, 20, 32),      # 20 rows
             (128, 64, 13, 24, 64),       # halves + odd image count + 24 rows
             (128, 256, 8, 32, 32)]       # four 64-channel blocks per parity (the stress configuration's width)


@pytest.mark.parametrize("fam", [pytest.param((3, 1e-6, b"<2, 1>"), id="fp32x2"), pytest.param((4, 1e-3, b"<1, 1>"), id="fp16")])
@pytest.mark.parametrize("shape", W4_SHAPES)
def test_wgrad4_halo(dev, shape, fam, monkeypatch):
    """Weight gradient of the 4x4 stride-2 layers through the halo kernel (wgrad3_halo_x3_kernel<NP, 1>: 2 x 2 taps over the
    four input parities, every operand row fetched and split once per workgroup; scaled-fp16 precisions): against torch
    in float64, fresh and accumulating, and against the generic split-once kernel (FCD_NO_WGRAD4=1) to the same bar."""
    from fcd_amd import _lib as L, ops
    O, I, n, h, w = shape
    prec, tol, tag = fam
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(41)
    big = torch.randn(n, I, 2 * h, 2 * w, generator=gen)
    small = torch.randn(n, O, h, w, generator=gen) * 0.01
    bd, sd = to_nhwc(big, g.IP).to(dev), to_nhwc(small, g.OP).to(dev)

    def desc(acc=False):
        d = g.wgrad_desc(n, 2 * h, 2 * w, precision=prec, accumulate=acc)
        return ops.set_wgrad_amax(d, ops.absmax(sd), ops.absmax(bd)) if prec >= 3 else d

    dw = ops.conv_wgrad(desc(), sd, bd)
    assert L.load().fcd_last_conv_kernel() == b"wgrad3_halo_x3_kernel" + tag
    ref = torch.nn.grad.conv2d_weight(big.double(), (O, I, 4, 4), small.double(), stride=2, padding=1)
    assert relerr(dw.cpu(), ref) <= tol
    base = float(ref.abs().max()) * 0.5
    acc = torch.full_like(dw, base)
    ops.conv_wgrad(desc(True), sd, bd, dw=acc)
    assert relerr(acc.cpu() - base, ref) <= 10 * tol
    monkeypatch.setenv("FCD_NO_WGRAD4", "1")
    old = ops.conv_wgrad(desc(), sd, bd)
    assert L.load().fcd_last_conv_kernel().startswith(b"conv_wgrad_x3p_kernel")
    assert relerr(old.cpu(), ref) <= tol and relerr(dw, old) <= 2 * tol


@pytest.mark.parametrize("fam", [pytest.param((3, 1e-6), id="fp32x2"), pytest.param((4, 1e-3), id="fp16")])
@pytest.mark.parametrize("geom", [(128, 64, 32, 64, 64), (128, 64, 33, 32, 32), (96, 40, 9, 64, 64), (128, 64, 36, 32, 64)])
def test_gather_phase_pairs_equal_single_phases(dev, geom, fam, monkeypatch):
    """The four 2 x 2-tap phases of a stride-2 adjoint with at most 64 output channels, two phases per workgroup on the
    256 x 128 tile's wave layout (template flag PP, scaled-fp16 precisions): same bits as one phase per workgroup on
    256 x 64 tiles (FCD_NO_PP=1), epilogue operands and fused InstanceNorm statistics included; within the precision's
    bar of torch in float64."""
    from fcd_amd import _lib as L, ops
    O, I, N, H, W = geom
    prec, tol = fam
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(57)
    wt = torch.randn(O, I, 4, 4, generator=gen) / (4 * I ** 0.5)
    wd = wt.to(dev)
    gy = torch.randn(N, O, H, W, generator=gen)
    mask = torch.randn(N, I, 2 * H, 2 * W, generator=gen)
    res = torch.randn(N, I, 2 * H, 2 * W, generator=gen)
    gd, md, rd = to_nhwc(gy, g.OP).to(dev), to_nhwc(mask, g.IP).to(dev), to_nhwc(res, g.IP).to(dev)
    wa, ga = ops.absmax(wd), ops.absmax(gd)

    def run():
        d = g.adj_desc(N, H, W, precision=prec, mask_slope=0.2)
        fmt = ops.weight_format(d)
        if fmt == 2:
            ops.set_amax(d, ga, wa)
        wp = ops.pack_weight_adj_h2(wd, g, wa) if fmt == 2 else ops.pack_weight_adj_x3(wd, g)
        out = ops.conv_gather(d, gd, wp, mask=md, residual=rd)
        k1 = L.load().fcd_last_conv_kernel()
        d2 = g.adj_desc(N, H, W, precision=prec)
        if fmt == 2:
            ops.set_amax(d2, ga, wa)
        z, mean, rstd, varu = ops.conv_gather_stats(d2, gd, wp, want_var=True)
        return out, z, mean, rstd, varu, k1

    new = run()
    assert b"(phase pairs)" in new[5], new[5]
    monkeypatch.setenv("FCD_NO_PP", "1")
    old = run()
    assert b"(phase pairs)" not in old[5] and b"<256, 64, 8, 1" in old[5], old[5]
    for a, b in zip(new[:5], old[:5]):          # outputs and statistics (one set of sums per 32 rows in both forms)
        assert torch.equal(a, b)
    ref = F.conv_transpose2d(gy.double(), wt.double(), None, 2, 1)
    assert relerr(from_nhwc(new[1].cpu(), I).double(), ref) <= tol
    ref2 = ref * torch.where(mask > 0, 1.0, 0.2) + res
    assert relerr(from_nhwc(new[0].cpu(), I).double(), ref2) <= tol
    assert relerr(new[2].cpu().double()[:, :I], ref.mean(dim=(2, 3))) <= 20 * tol


@pytest.mark.parametrize("shape", [(2048, 1024, 8, 4, 4, 4), (512, 256, 4, 2, 16, 16), (128, 64, 2, 2, 32, 32), (72, 20, 2, 3, 8, 8),
                                   (1024, 512, 32, 16, 8, 8)])
def test_wgrad_two_operand_pairs(dev, shape):
    """fcd_conv_wgrad2: the split-once kernel's K loop over two operand pairs of one geometry (the discriminator's paired
    batch and its gradient-penalty batch) -- one pass over dw, fresh and accumulating, direct and split-K shapes, a ragged
    last K-tile in the second pair -- against torch in float64 and against two accumulating single-pair calls."""
    from fcd_amd import ops
    O, I, n1, n2, h, w = shape
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(61)
    xs = [torch.randn(n, I, h, w, generator=gen) for n in (n1, n2)]
    sh, sw = g.small_hw(h, w)
    gys = [torch.randn(n, O, sh, sw, generator=gen) * s for n, s in ((n1, 1.0), (n2, 3.0))]
    xd = [to_nhwc(x, g.IP).to(dev) for x in xs]
    gd = [to_nhwc(gy, g.OP).to(dev) for gy in gys]
    d = g.wgrad_desc(n1, h, w, precision=2)
    assert ops.conv_wgrad2_supported(d)
    got = ops.conv_wgrad2(d, gd[0], xd[0], gd[1], xd[1])
    ref = sum(torch.nn.grad.conv2d_weight(x.double(), (O, I, 4, 4), gy.double(), stride=2, padding=1) for x, gy in zip(xs, gys))
    assert relerr(got, ref) <= 1e-5
    two = ops.conv_wgrad(g.wgrad_desc(n1, h, w, precision=2), gd[0], xd[0])
    ops.conv_wgrad(g.wgrad_desc(n2, h, w, precision=2, accumulate=True), gd[1], xd[1], dw=two)
    assert relerr(got, two) <= 2e-6
    acc = torch.full_like(got, 0.125)
    ops.conv_wgrad2(g.wgrad_desc(n1, h, w, precision=2, accumulate=True), gd[0], xd[0], gd[1], xd[1], dw=acc)
    assert relerr(acc - 0.125, ref) <= 1e-4
    # shapes and precisions without the two-segment form say so
    assert not ops.conv_wgrad2_supported(g.wgrad_desc(n1, h, w, precision=0))
    assert not ops.conv_wgrad2_supported(ops.ConvGeom(16, I, 4, 2, 1).wgrad_desc(n1, h, w, precision=2))


# (kernel size, O, I, n1, n2, small h, small w): shapes whose two-pair weight gradient takes the halo kernel; image counts
# chosen so that a split's image range straddles the boundary between the pairs
W3_PAIR_SHAPES = [(3, 128, 128, 100, 100, 8, 32),     # 3 x 3: three images per split, split 33 = images 99, 100 (pair 2's first), 101
                  (3, 256, 256, 5, 11, 32, 32),       # the residual blocks' geometry, one image per split, uneven pairs
                  (4, 128, 64, 31, 21, 8, 32),        # 4 x 4 stride 2: two images per split, an odd first pair
                  (4, 64, 64, 40, 41, 8, 64),         # 64-pixel gradient rows: three 32-column halves per split
                  (4, 128, 64, 9, 7, 8, 16)]          # 16-pixel rows: no halo form, the split-once kernel's two segments in every precision


@pytest.mark.parametrize("fam", [pytest.param((2, 2e-6, b""), id="fp32x3"), pytest.param((3, 1e-6, b"<2"), id="fp32x2"),
                                 pytest.param((4, 1e-3, b"<1"), id="fp16")])
@pytest.mark.parametrize("shape", W3_PAIR_SHAPES)
def test_wgrad_two_operand_pairs_halo(dev, shape, fam):
    """fcd_conv_wgrad2 on the halo kernel (3 x 3 stride 1 and 4 x 4 stride 2; the generator's two backward passes of a G
    step): the second pair's images follow the first's in the splits' image ranges, the scaled-fp16 forms take one scale
    per operand from the larger of the two pairs' maxima (the pairs differ by 8x and 1/4x here).  Against torch in float64
    and against two accumulating single-pair calls."""
    from fcd_amd import _lib as L, ops
    k, O, I, n1, n2, h, w = shape
    prec, tol, tag = fam
    g = ops.ConvGeom(O, I, k, 1, 1) if k == 3 else ops.ConvGeom(O, I, 4, 2, 1)
    s = 1 if k == 3 else 2
    gen = torch.Generator().manual_seed(67)
    xs = [torch.randn(n, I, s * h, s * w, generator=gen) * sc for n, sc in ((n1, 1.0), (n2, 0.25))]
    gys = [torch.randn(n, O, h, w, generator=gen) * sc for n, sc in ((n1, 0.01), (n2, 0.08))]
    xd = [to_nhwc(x, g.IP).to(dev) for x in xs]
    gd = [to_nhwc(gy, g.OP).to(dev) for gy in gys]

    def desc(n, j, acc=False):
        d = g.wgrad_desc(n, s * h, s * w, precision=prec, accumulate=acc)
        return ops.set_wgrad_amax(d, ops.absmax(gd[j]), ops.absmax(xd[j])) if prec >= 3 else d

    mx2 = (ops.absmax(gd[1]), ops.absmax(xd[1])) if prec >= 3 else None
    d = desc(n1, 0)
    assert ops.conv_wgrad2_supported(d)
    got = ops.conv_wgrad2(d, gd[0], xd[0], gd[1], xd[1], amax2=mx2)
    kern = L.load().fcd_last_conv_kernel()
    if w >= 32 and (k == 3 or prec >= 3):
        assert kern.startswith(b"wgrad3_halo_x3_kernel" + tag), kern
    else:
        # (4 x 4 under bf16 x 3, or rows the halo kernel does not take: the split-once kernel's two segments)
        assert kern.startswith(b"conv_wgrad_x3p_kernel") and kern.endswith({2: b"3>", 3: b"2>", 4: b"1>"}[prec]), kern
    ref = sum(torch.nn.grad.conv2d_weight(x.double(), (O, I, k, k), gy.double(), stride=s, padding=1) for x, gy in zip(xs, gys))
    assert relerr(got.cpu(), ref) <= tol
    two = ops.conv_wgrad(desc(n1, 0), gd[0], xd[0])
    ops.conv_wgrad(desc(n2, 1, True), gd[1], xd[1], dw=two)
    assert relerr(got, two) <= 2 * tol
    base = float(ref.abs().max()) * 0.5
    acc = torch.full_like(got, base)
    ops.conv_wgrad2(desc(n1, 0, True), gd[0], xd[0], gd[1], xd[1], dw=acc, amax2=mx2)
    assert relerr(acc.cpu() - base, ref) <= 10 * tol
    if prec >= 3:
        # without the second pair's maxima there is no common scale: the bf16 x 3 form of the same kernels
        alt = ops.conv_wgrad2(desc(n1, 0), gd[0], xd[0], gd[1], xd[1])
        assert relerr(alt.cpu(), ref) <= 2e-6
        ops.check_overflow(dev)


# (O, I, N, small h, small w): 4 x 4 stride-2 pad-1 forward convolutions that take the space-to-depth A-halo form
S2D_GEOMS = [(128, 64, 32, 64, 64),       # the generator's 64 -> 128 down-sampling layer at a paired call's 32 samples
             (256, 128, 32, 32, 32),      # its 128 -> 256 layer: two column tiles
             (160, 96, 17, 64, 64),       # ragged output channels (the second column tile is a quarter full), 3 slices per parity
             (128, 32, 89, 24, 32)]       # one slice per parity, 24 rows per image (three row tiles), an odd tile count (267)


@pytest.mark.parametrize("fam", [pytest.param((2, 2e-6, b"3>"), id="fp32x3"), pytest.param((3, 1e-6, b"2>"), id="fp32x2"),
                                 pytest.param((4, 1e-3, b"1>"), id="fp16")])
@pytest.mark.parametrize("geom", S2D_GEOMS)
def test_gather_space_to_depth(dev, geom, fam, monkeypatch):
    """4 x 4 stride-2 pad-1 convolutions through the A-halo kernel's space-to-depth form (template flag S2D: 2 x 2 taps over
    the input's four parities as 4 Cin virtual channels; a 32-channel slice of one parity fetched and split once per
    workgroup): against torch in float64 with bias + LeakyReLU in the epilogue, the fused InstanceNorm statistics of the
    plain convolution against float64, and everything against the per-tap kernel (FCD_NO_S2D=1; another K order, so to
    rounding)."""
    from fcd_amd import _lib as L, ops
    O, I, N, h, w = geom
    prec, tol, tag = fam
    g = ops.ConvGeom(O, I, 4, 2, 1)
    gen = torch.Generator().manual_seed(71)
    wt = torch.randn(O, I, 4, 4, generator=gen) / (4 * I ** 0.5)
    x = torch.randn(N, I, 2 * h, 2 * w, generator=gen)
    x.view(-1)[::50021] *= 300.0
    b = torch.randn(O, generator=gen)
    wd, xd = wt.to(dev), to_nhwc(x, g.IP).to(dev)
    wa, xa = ops.absmax(wd), ops.absmax(xd)

    def desc(**kw):
        d = g.fwd_desc(N, 2 * h, 2 * w, precision=prec, **kw)
        return ops.set_amax(d, xa, wa) if prec >= 3 else d

    fmt = ops.weight_format(desc())
    assert fmt == (2 if prec >= 3 else 1)
    wp = ops.pack_weight_fwd_h2(wd, g, wa) if fmt == 2 else ops.pack_weight_fwd_x3(wd, g)

    def run():
        out = ops.conv_gather(desc(act_slope=0.01), xd, wp, bias=b.to(dev))
        k = L.load().fcd_last_conv_kernel()
        z, mean, rstd, varu = ops.conv_gather_stats(desc(), xd, wp, want_var=True)
        return out, z, mean, rstd, varu, k

    new = run()
    assert b"(space to depth)" in new[5] and new[5].endswith(b"true, " + tag[:1] + b">"), new[5]
    monkeypatch.setenv("FCD_NO_S2D", "1")
    old = run()
    assert b"(space to depth)" not in old[5], old[5]
    ref = F.conv2d(x.double(), wt.double(), None, 2, 1)
    assert relerr(from_nhwc(new[0].cpu(), O).double(), F.leaky_relu(ref + b.double().view(1, -1, 1, 1), 0.01)) <= tol
    assert relerr(from_nhwc(new[1].cpu(), O).double(), ref) <= tol
    assert relerr(new[2].cpu().double()[:, :O], ref.mean(dim=(2, 3))) <= 20 * tol
    assert relerr(new[4].cpu().double()[:, :O], ref.var(dim=(2, 3), unbiased=True)) <= 20 * tol
    for a_, b_ in zip(new[:5], old[:5]):
        assert relerr(a_, b_) <= 2 * tol
    if prec >= 3:
        ops.check_overflow(dev)
