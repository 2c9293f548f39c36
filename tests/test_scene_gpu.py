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
