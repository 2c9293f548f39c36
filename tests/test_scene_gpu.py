"""generate_masks on a tiled scene: HIP path vs the reference-made fixture and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import fcd_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def tiny_solver(dev):
    from fcd_amd import FCDSolver
    s = FCDSolver(None, c_dim=1, image_size=32, g_conv_dim=8, d_conv_dim=8, g_repeat_num=2, d_repeat_num=3,
                  num_channels=10, batch_size=2)
    s.G.load_state_dict(O.synth_generator_state(8, 1, 2, 10))
    return s.to(dev)


def test_predict_scene_matches_reference_fixture(dev):
    from fcd_amd.scene import generate_scene_mask, predict_scene
    z = np.load(os.path.join(GOLD, "tiny_scene.npz"))
    s = tiny_solver(dev)
    s.G.eval()
    diff = predict_scene(z["scene"], s, batch_size=2)
    assert diff.dtype == np.float64 and diff.shape == z["diff"].shape
    assert np.abs(diff - z["diff"]).max() <= 1e-5
    far = np.abs(z["diff"] - 0.02) > 1e-5
    assert np.array_equal(s.binarize(diff, 0.02)[far], z["mask_0.02"][far])
    d2, m2 = generate_scene_mask(z["scene"], s, 0.02, batch_size=2)
    assert m2.dtype == np.uint8 and bool((m2[:9] == 0).all())     # nodata border stays clear/invalid


def test_pseudo_masks_encoding(dev):
    s = tiny_solver(dev)
    batch, _ = O.synth_batch(4, 32, 10)
    x = batch[0]
    valid = torch.ones(4, 32, 32, dtype=torch.uint8)
    valid[x[:, 0] == -1] = 0
    samples = [{"image": x[:2], "mask": valid[:2]}, {"image": x[2:], "mask": valid[2:]}]
    m = s.make_psuedo_masks(0.02794, samples)
    assert m.shape == (4, 32, 32) and m.dtype == np.uint8
    assert set(np.unique(m).tolist()) <= {0, 1, 2} and bool((m[valid.numpy() == 0] == 0).all())
    d = O.difference_map(O.synth_generator_state(8, 1, 2, 10), x, False)
    ref = O.pseudo_mask(d, 0.02794, valid > 0).numpy()
    far = (d - 0.02794).abs().numpy() > 1e-5
    assert np.array_equal(m[far], ref[far])
    assert s.generate_masks.__func__ is s.make_psuedo_masks.__func__
    assert s.pseudo_mask_cm.shape == (2, 2) and int(s.pseudo_mask_cm.sum()) == int(valid.sum())
    assert int(s.pseudo_mask_cm[1].sum()) == 0                      # every valid pixel of this target is "clear"


def _scene(h, w, seed=5):
    rng = np.random.default_rng(seed)
    sc = rng.integers(1, 65536, size=(h, w, 10), dtype=np.uint16)
    sc[:9] = 0                       # nodata border (all bands zero)
    sc[:, -7:] = 0
    sc[40:80, 0:40] = 0              # one patch that is entirely nodata
    return sc


@pytest.mark.parametrize("hw", [(100, 77), (96, 64), (33, 130)])
def test_device_scene_path_equals_host_path(dev, hw):
    """Tiling (reflect padding, ragged and exact multiples of the patch), normalisation, the
    all-nodata batch skip and the float64 stitch on the device are bit-identical to the
    reference's host-side numpy tiling around the same G."""
    from fcd_amd.scene import generate_scene_mask, predict_scene, predict_scene_host
    s = tiny_solver(dev)
    s.G.eval()
    sc = _scene(*hw)
    for bs in (1, 3):
        a = predict_scene(sc, s, batch_size=bs)
        b = predict_scene_host(sc, s, batch_size=bs)
        assert a.dtype == np.float64 and np.array_equal(a, b)
    # nodata zeroing + float64 compare on the device == numpy on the host raster
    d, m = generate_scene_mask(sc, s, 0.02, batch_size=3)
    ref = predict_scene_host(sc, s, batch_size=3)
    ref[~np.any(sc != 0, axis=2)] = 0
    assert np.array_equal(d, ref) and np.array_equal(m, (ref > 0.02).astype(np.uint8))
    # a threshold that equals a raster value exactly: strict '>' in float64
    thr = float(ref[ref > 0].flat[0])
    _, m2 = generate_scene_mask(sc, s, thr, batch_size=3)
    assert np.array_equal(m2, (ref > thr).astype(np.uint8))


def test_scene_tiles_kernel_against_numpy(dev):
    """fcd_scene_tiles alone: numpy.pad(..., 'reflect') + slicing + normalisation, bit-exact."""
    from fcd_amd import ops
    from fcd_amd.scene import scene_patches
    sc = _scene(70, 45, seed=9)
    patch = 32
    origins, pad_h, pad_w, _ = scene_patches(70, 45, patch, 0)
    padded = np.pad(sc, (pad_h, pad_w, (0, 0)), "reflect")
    ref = np.stack([O.normalize_u16(padded[r:r + patch, c:c + patch]) for r, c in origins]).transpose(0, 3, 1, 2)
    x, flag = ops.scene_tiles(torch.from_numpy(sc).to(dev), torch.tensor(origins, dtype=torch.int32, device=dev), patch)
    assert int(flag.item()) == 1 and np.array_equal(x.cpu().numpy(), ref)
    z = np.zeros((40, 40, 10), dtype=np.uint16)
    x, flag = ops.scene_tiles(torch.from_numpy(z).to(dev), torch.tensor([[0, 0]], dtype=torch.int32, device=dev), patch)
    assert int(flag.item()) == 0 and bool((x == -1).all())
    with pytest.raises(RuntimeError):       # scene smaller than a patch: multi-reflection is not supported
        ops.scene_tiles(torch.from_numpy(z[:20]).to(dev), torch.tensor([[0, 0]], dtype=torch.int32, device=dev), patch)


def test_find_best_threshold_on_device(dev):
    """Device min/max + confusion histogram reproduce the oracle's numpy sweep exactly
    (integer counts, same fp32 comparison), incl. invalid pixels and ties at the thresholds."""
    from fcd_amd import ops
    from fcd_amd.solver import best_threshold_from_hist
    s = tiny_solver(dev)
    batch, _ = O.synth_batch(6, 32, 10)
    x = batch[0]
    rng = np.random.default_rng(3)
    samples = []
    for i in range(0, 6, 2):
        d = O.difference_map(O.synth_generator_state(8, 1, 2, 10), x[i:i + 2], False).numpy()
        tgt = (d > np.median(d)).astype(np.uint8) + 1            # cloudy where the map is high ...
        flip = rng.random(tgt.shape) < 0.2
        tgt[flip] = 3 - tgt[flip]                                # ... with label noise
        tgt[x[i:i + 2, 0].numpy() == -1] = 0                     # nodata is invalid
        samples.append({"image": x[i:i + 2], "mask": torch.from_numpy(tgt)})
    thr, _ = s.find_best_threshold(n_thresholds=30, dataset=samples)
    # oracle on the HIP path's own difference maps (the sweep is what is under test here)
    s.G.eval()
    preds, tgts = [], []
    for smp in samples:
        d = s.compute_difference_map(smp["image"].to(dev)).cpu().numpy()
        t = smp["mask"].numpy()
        preds.append(d[t > 0])
        tgts.append(t[t > 0] - 1)
    preds, tgts = np.concatenate(preds), np.concatenate(tgts)
    assert thr == O.find_best_threshold(preds, tgts, 30)
    # the histogram itself, against numpy counts for every threshold (no early break)
    thresholds = np.linspace(preds.min(), preds.max(), 30)
    thresholds[7] = preds[11]                                    # a threshold that ties with a value
    thresholds = np.sort(thresholds)
    hist = torch.zeros((2, 31), dtype=torch.int64, device=dev)
    dd = torch.from_numpy(np.concatenate([p for p in preds.reshape(1, -1)])).to(dev)
    tt = torch.from_numpy((tgts + 1).astype(np.uint8)).to(dev)
    ops.threshold_hist(dd, tt, torch.from_numpy(thresholds.astype(np.float32)).to(dev), hist)
    h = hist.cpu().numpy()
    for t, th in enumerate(thresholds):
        p = preds > np.float32(th)
        for cls in (0, 1):
            assert h[cls, t + 1:].sum() == int(np.sum(p & (tgts == cls))), (t, cls)
    assert h.sum() == preds.size
    mm = torch.tensor([float("inf"), 0.0], device=dev)
    ops.valid_minmax(dd, tt, mm)
    assert float(mm[0]) == float(preds.min()) and float(mm[1]) == float(preds.max())
    assert best_threshold_from_hist(h, thresholds) is not None


def test_validation_step_confusion_matrix(dev):
    """validation_step / validation_epoch_end (fcd_solver.py:497-544): device confusion counts equal
    numpy's on the same difference maps; metrics as metrics.py computes them."""
    s = tiny_solver(dev)
    s.G.eval()
    s.threshold = 0.05
    batch, _ = O.synth_batch(4, 32, 10)
    x = batch[0]
    rng = np.random.default_rng(11)
    tgt = rng.integers(0, 3, size=(4, 32, 32)).astype(np.uint8)
    cm_ref = np.zeros((2, 2), dtype=np.int64)
    for i in (0, 2):
        xb, tb = x[i:i + 2], torch.from_numpy(tgt[i:i + 2])
        s.validation_step((xb, None, None, None, None, tb))
        d = s.compute_difference_map(xb.to(dev)).cpu().numpy()
        pred = (d > np.float32(s.threshold)).astype(np.uint8)
        valid = tgt[i:i + 2] > 0
        cm_ref += O.compute_confusion_matrix(pred[valid], tgt[i:i + 2][valid] - 1, 2)
    out = s.validation_epoch_end()
    assert np.array_equal(out["confusion_matrix"], cm_ref) and cm_ref.sum() == int((tgt > 0).sum())
    assert out["F1"] == O.f1_score(cm_ref)
    assert abs(out["Accuracy"] - np.diag(cm_ref).sum() / cm_ref.sum()) < 1e-15
    assert s.validation_epoch_end()["confusion_matrix"].sum() == 0      # accumulator was reset


def test_scene_at_stated_size_device_equals_host(dev):
    """BASELINE.json configs[3] at its stated size: a 2000 x 1900 x 10 synthetic scene (not a multiple of
    the patch: reflect padding on both edges), 128x128 patches, batch 64, the full-size generator.
    Device path == host (reference-style numpy tiling / stitching) path bit for bit, the mask == numpy's
    float64 '>' on the raster, all-nodata batches are skipped on both sides (the first batch of 64 tiles
    is nodata), and the write_tifs byte rasters equal numpy's."""
    from fcd_amd import FCDSolver, ops
    from fcd_amd.scene import generate_scene_mask, predict_scene_host, scene_patches
    h, w = 2000, 1900
    rng = np.random.default_rng(4)
    sc = rng.integers(1, 65536, size=(h, w, 10), dtype=np.uint16)
    sc[:700] = 0                      # more than the first 64 tiles (4.27 tile rows): batch 0 is skipped
    sc[:, :48] = 0
    sc[-30:] = 0
    s = FCDSolver(None, precision="fp32x3")
    s.G.load_state_dict(O.synth_generator_state())
    s.to(dev)
    s.G.eval()
    # per-tile nodata flags against numpy on the reflect-padded scene
    origins, pad_h, pad_w, _ = scene_patches(h, w, 128, 0)
    assert len(origins) == 16 * 15
    padded = np.pad(sc, (pad_h, pad_w, (0, 0)), "reflect")
    ref_valid = np.array([padded[r:r + 128, c:c + 128].any() for r, c in origins], dtype=np.uint8)
    tv = ops.scene_tile_valid(torch.from_numpy(sc).to(dev), torch.tensor(origins, dtype=torch.int32, device=dev), 128)
    assert np.array_equal(tv.cpu().numpy(), ref_valid) and not ref_valid[:64].any() and ref_valid[64:].any()
    calls = []
    orig = s.compute_difference_map
    s.compute_difference_map = lambda x: (calls.append(x.shape[0]), orig(x))[1]
    d, m, du8, mu8 = generate_scene_mask(sc, s, 0.02, batch_size=64, tif_bytes=True)
    assert calls == [64, 64, 48]      # 240 tiles in batches of 64, the first one skipped
    s.compute_difference_map = orig
    ref = predict_scene_host(sc, s, batch_size=64)
    valid = np.any(sc != 0, axis=2)
    ref[~valid] = 0
    assert d.dtype == np.float64 and np.array_equal(d, ref)
    assert np.array_equal(m, (ref > 0.02).astype(np.uint8)) and 0 < int(m.sum()) < valid.sum()
    # evaluate.py:384-390
    assert np.array_equal(du8, (ref * 255).astype(np.uint8))
    tif = m.copy()
    tif[tif == 0] = 128
    tif[tif == 1] = 255
    tif[~valid] = 0
    assert np.array_equal(mu8, tif) and set(np.unique(mu8).tolist()) == {0, 128, 255}
    # an explicit validity mask (the reference passes target > 0)
    vm = valid.copy()
    vm[1000:1100] = False
    _, _, _, mu8b = generate_scene_mask(sc, s, 0.02, batch_size=64, tif_bytes=True, valid_mask=vm)
    tif[~vm] = 0
    assert np.array_equal(mu8b, tif)


def test_pseudo_mask_kernel_counts_and_encoding(dev):
    """fcd_pseudo_mask: threshold + confusion counts + {0, 1, 2} encoding in one pass == numpy
    (fcd_solver.py:762-772), incl. values exactly at fp32(thr)."""
    from fcd_amd import ops
    rng = np.random.default_rng(21)
    n = 3 * 128 * 128 + 17
    thr = 0.02794
    d = rng.random(n).astype(np.float32) * 0.06
    t32 = np.float32(thr)
    d[:6] = [t32, np.nextafter(t32, np.float32(1)), np.nextafter(t32, np.float32(0)), 0.0, 1.0, t32]
    tgt = rng.integers(0, 3, size=n).astype(np.uint8)
    cm = torch.zeros((2, 2), dtype=torch.int64, device=dev)
    out = ops.pseudo_mask(torch.from_numpy(d).to(dev), torch.from_numpy(tgt).to(dev), thr, cm=cm)
    pred = (torch.from_numpy(d) > thr).numpy().astype(np.uint8)          # torch's fp32 compare
    ref = pred + 1
    ref[tgt == 0] = 0
    assert np.array_equal(out.cpu().numpy(), ref)
    valid = tgt > 0
    cm_ref = O.compute_confusion_matrix(pred[valid], tgt[valid] - 1, 2)
    assert np.array_equal(cm.cpu().numpy(), cm_ref.astype(np.int64))
    ops.pseudo_mask(torch.from_numpy(d).to(dev), torch.from_numpy(tgt).to(dev), thr, cm=cm)    # accumulates
    assert np.array_equal(cm.cpu().numpy(), 2 * cm_ref.astype(np.int64))
    out2 = ops.pseudo_mask(torch.from_numpy(d).to(dev), None, thr)                                   # no target: all valid
    assert np.array_equal(out2.cpu().numpy(), pred + 1)


@pytest.mark.parametrize("mode", ["fp32", "fp32x3", "fp32x2"])
def test_eval_difference_map_full_width(dev, mode):
    """The inference path at the baseline's width (conv_dim 64, 128x128x10, eval mode: InstanceNorm folded into the
    convolutions; under fp32x2 the folded weights are two fp16 planes and every layer's epilogue leaves max |output|
    for the next layer's operand scale): the difference map of a batch against the float64 oracle, and the mask
    against the oracle's away from the threshold."""
    from fcd_amd import FCDSolver
    gs = O.synth_generator_state()
    (x, _, _, _, _), _ = O.synth_batch(2, 128, 10, seed=21)
    ref = O.difference_map(O.to_dtype(gs, torch.float64), x.double(), training=False)
    s = FCDSolver(None, precision=mode)
    s.G.load_state_dict(gs)
    s.to(dev)
    s.G.eval()
    diff, mask = s.difference_and_mask(x.to(dev), 0.02)
    err = float((diff.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-6, err
    far = (ref - 0.02).abs() > 1e-5
    assert bool((mask.cpu()[far] == (ref > 0.02).to(torch.uint8)[far]).all())
    # a larger batch takes other tile shapes (256-row tiles, A-halo form): same values per sample
    xb = x.repeat(16, 1, 1, 1).to(dev)
    diff32, _ = s.difference_and_mask(xb, 0.02)
    assert float((diff32[:2] - diff).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert float((diff32[30:32] - diff).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_weight_maxima_pool_follows_the_weights(dev):
    """The generator's weight maxima come from one batched launch per set of parameter versions (WeightMaxima): after an
    in-place update of one weight, of all weights, after load_state_dict and for a deep copy of the network the fp32x2
    forward equals the forward of a freshly built network with the same weights, bit for bit."""
    import copy
    from fcd_amd import FCDSolver
    gs = O.synth_generator_state()
    (x, _, _, _, _), _ = O.synth_batch(2, 128, 10, seed=5)
    xd = x.to(dev)
    c = torch.zeros(2, 1, device=dev)

    def fresh(state):
        s2 = FCDSolver(None, precision="fp32x2")
        s2.G.load_state_dict(state)
        s2.to(dev)
        with torch.no_grad():
            return s2.G(xd, c)

    s = FCDSolver(None, precision="fp32x2")
    s.G.load_state_dict(gs)
    s.to(dev)
    pool = s.G._wmax
    with torch.no_grad():
        y0 = s.G(xd, c)
        assert pool.keys is not None and torch.equal(y0, fresh(gs))
        for r, cv in zip(pool.buf, pool.convs):
            assert float(r.max()) == float(cv.weight.abs().max())
        # one weight changes in place (its version moves): the pool is refreshed at the next forward
        s.G.main[9].main[0].weight.mul_(1.5)
        y1 = s.G(xd, c)
        st = {k: v.detach().cpu().clone() for k, v in s.G.state_dict().items()}
        assert not torch.equal(y1, y0) and torch.equal(y1, fresh(st))
        # a weight changes between the refresh and its use: that convolution falls back to its own maximum
        cv = pool.convs[3]
        cv.weight.mul_(0.5)
        assert pool.row(cv, cv._key()) is None
        assert float(cv.weight_amax().max()) == float(cv.weight.abs().max())
        # every weight, then a state dict
        for p in s.G.parameters():
            p.mul_(1.01)
        st = {k: v.detach().cpu().clone() for k, v in s.G.state_dict().items()}
        assert torch.equal(s.G(xd, c), fresh(st))
        s.G.load_state_dict(gs)
        assert torch.equal(s.G(xd, c), y0)
        g2 = copy.deepcopy(s.G)
        assert torch.equal(g2(xd, c), y0)
        g2.main[0].weight.mul_(2.0)
        st = {k: v.detach().cpu().clone() for k, v in g2.state_dict().items()}
        assert torch.equal(g2(xd, c), fresh(st))
        assert torch.equal(s.G(xd, c), y0)
