"""Input pipeline on the device against the numpy oracle: bit-exact.

The normalisation formula itself is pinned by the reference-generated scene fixture
(tests/golden/tiny_scene.npz, test_oracle_golden); the flips / shifts are index arithmetic.
albumentations is absent, so the augmentation chain has no reference-side fixture: the numpy
slicing in oracle.patch_batch is its published definition (``img[:, ::-1]``, ``img[::-1]``).
"""
import numpy as np
import pytest
import torch

from oracle import fcd_oracle as O

pytestmark = pytest.mark.gpu


def _pool(n=7, h=24, w=20, c=10, seed=3):
    rng = np.random.default_rng(seed)
    images = rng.integers(0, 65536, size=(n, h, w, c), dtype=np.uint16)
    images[1, :5, :7, :] = 0
    masks = rng.integers(0, 3, size=(n, h, w), dtype=np.uint8)
    labels = rng.integers(0, 2, size=(n,), dtype=np.uint8)
    return images, masks, labels


@pytest.mark.parametrize("with_masks", [False, True])
def test_patch_batch_bit_exact(with_masks):
    from fcd_amd import ops
    images, masks, _ = _pool()
    index = np.array([3, 0, 6, 1, 1, 5], dtype=np.int64)
    aug = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 4, -3], [0, -23, 19], [1, 24, 0]], dtype=np.int32)
    want_x, want_m = O.patch_batch(images, index, aug, masks if with_masks else None)
    dev = "cuda"
    x, m = ops.patch_batch(torch.from_numpy(images).to(dev), torch.from_numpy(index).to(dev),
                           torch.from_numpy(aug).to(dev), torch.from_numpy(masks).to(dev) if with_masks else None)
    assert np.array_equal(x.cpu().numpy().view(np.uint32), want_x.view(np.uint32))
    if with_masks:
        assert np.array_equal(m.cpu().numpy(), want_m)
    else:
        assert m is None
    # no augmentation argument == all-zero augmentation
    x0, _ = ops.patch_batch(torch.from_numpy(images).to(dev), torch.from_numpy(index).to(dev))
    want0, _ = O.patch_batch(images, index)
    assert np.array_equal(x0.cpu().numpy().view(np.uint32), want0.view(np.uint32))
    # nodata survives: uint16 0 -> exactly -1.0
    assert (x0[3, :, :5, :7] == -1.0).all()


def test_pool_and_loader():
    from fcd_amd.data_loader import L8BiomePatchPool, PatchLoader, on_before_batch_transfer
    images, masks, labels = _pool(n=11)
    pool = L8BiomePatchPool(images, labels, masks, shuffle=True, seed=5)
    ref_idx = np.arange(11)
    np.random.default_rng(5).shuffle(ref_idx)                      # data_loader.py:242-243
    assert np.array_equal(pool.indices.cpu().numpy(), ref_idx)
    # validation loader: doubled batch, no augmentation, masks, pool order
    ld = PatchLoader(pool, batch_size=2, mode="val", ret_mask=True)
    assert len(ld) == 3
    seen = []
    for bi, batch in enumerate(ld):
        pos = np.arange(bi * 4, min(bi * 4 + 4, 11))
        want_x, want_m = O.patch_batch(images, ref_idx[pos], None, masks)
        assert np.array_equal(batch["image"].cpu().numpy(), want_x)
        assert np.array_equal(batch["mask"].cpu().numpy(), want_m)
        assert np.array_equal(batch["label"].cpu().numpy(), labels[ref_idx[pos]].astype(np.float32))
        seen.append(batch["image"].shape[0])
    assert seen == [4, 4, 3]
    # training loader: every sample once per epoch, flips drawn from the generator
    g = torch.Generator().manual_seed(1)
    tl = PatchLoader(pool, batch_size=4, mode="train", generator=g, max_shift=3)
    g2 = torch.Generator().manual_seed(1)
    order = torch.randperm(11, generator=g2)
    for bi, batch in enumerate(tl):
        pos = order[bi * 4: bi * 4 + 4]
        b = pos.numel()
        flags = (torch.rand(b, generator=g2) < 0.5).int() + 2 * (torch.rand(b, generator=g2) < 0.5).int()
        shift = torch.randint(-3, 4, (b, 2), generator=g2, dtype=torch.int32)
        aug = torch.cat([flags.view(b, 1), shift], 1).numpy()
        want_x, _ = O.patch_batch(images, ref_idx[pos.numpy()], aug)
        assert np.array_equal(batch["image"].cpu().numpy(), want_x)
        tup = on_before_batch_transfer(batch, generator=torch.Generator().manual_seed(9))
        x_real, c_org, c_trg, label_org, label_trg = tup
        perm = torch.randperm(b, generator=torch.Generator().manual_seed(9))
        assert label_org.shape == (b, 1) and torch.equal(label_trg.cpu(), label_org.cpu()[perm])
        assert torch.equal(c_org, label_org) and c_org.data_ptr() != label_org.data_ptr()
        assert torch.equal(c_trg, label_trg)
    # two ranks split an epoch without overlap (DistributedSampler-style)
    a = PatchLoader(pool, 4, "val", rank=0, world_size=2)
    b2 = PatchLoader(pool, 4, "val", rank=1, world_size=2)
    na = sum(x["image"].shape[0] for x in a)
    nb = sum(x["image"].shape[0] for x in b2)
    assert na == nb == 6


def test_pool_rejects_bad_positions():
    from fcd_amd.data_loader import L8BiomePatchPool
    images, masks, labels = _pool()
    pool = L8BiomePatchPool(images, labels, only_cloudy=True)
    assert len(pool) == int((labels == 1).sum())
    with pytest.raises(IndexError):
        pool.gather(torch.tensor([len(pool)]))
    with pytest.raises(RuntimeError):
        pool.gather(torch.tensor([0]), ret_mask=True)
    # device-resident positions: no device-side assert (it would poison the process); the bad position is clamped for
    # the gather and reported by check_positions()
    out = pool.gather(torch.tensor([0, len(pool) + 3], device=pool.device))
    assert out["image"].shape[0] == 2 and bool(torch.isfinite(out["image"]).all())
    with pytest.raises(IndexError):
        pool.check_positions()
    pool.gather(torch.tensor([0, len(pool) - 1], device=pool.device))
    pool.check_positions()          # in range: nothing raised, flag cleared


def test_loader_feeds_train_and_validation_steps():
    """End to end: patch pool -> PatchLoader -> on_before_batch_transfer -> train_step, then the
    validation loader (masks alongside) -> validation_step / validation_epoch_end."""
    from fcd_amd import FCDSolver
    from fcd_amd.data_loader import L8BiomePatchPool, PatchLoader, on_before_batch_transfer
    rng = np.random.default_rng(1)
    n, size = 12, 32
    images = rng.integers(1, 65536, size=(n, size, size, 10), dtype=np.uint16)
    images[::3, :6, :9, :] = 0
    masks = rng.integers(0, 3, size=(n, size, size), dtype=np.uint8)
    labels = (masks == 2).any(axis=(1, 2)).astype(np.uint8)          # prepare_landsat8_biome.py:170
    pool = L8BiomePatchPool(images, labels, masks)
    s = FCDSolver(None, image_size=size, g_conv_dim=8, d_conv_dim=8, g_repeat_num=2, d_repeat_num=3, batch_size=4,
                  precision="fp32x3").to("cuda")
    w0 = s.G.main[0].weight.detach().clone()
    g = torch.Generator().manual_seed(0)
    steps = 0
    for batch in PatchLoader(pool, batch_size=4, mode="train", generator=g, max_shift=2):
        tup = on_before_batch_transfer(batch, generator=g)
        assert tup[0].shape == (4, 10, size, size) and tup[1].shape == (4, 1)
        d_loss, g_loss = s.train_step(tup, force_g=True)
        assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
        steps += 1
    assert steps == 3 and float((s.G.main[0].weight - w0).abs().max()) > 0
    s.G.eval()
    seen = 0
    for batch in PatchLoader(pool, batch_size=4, mode="val", ret_mask=True):
        tup = on_before_batch_transfer(batch)
        assert len(tup) == 6 and tup[5].dtype == torch.uint8
        s.validation_step(tup)
        seen += tup[0].shape[0]
    res = s.validation_epoch_end()
    assert seen == n and int(res["confusion_matrix"].sum()) == int((masks > 0).sum())
    assert 0.0 <= res["F1"] <= 1.0
