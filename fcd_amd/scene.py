"""Scene-level mask generation: tile a Landsat-8 scene, run G towards "clear", stitch.

Mirrors ``evaluate.predict_scene`` (``evaluate.py:84-123``) with ``PatchDataset``'s tiling
(``data_loader.py:74-103``, crop_size 0): reflect-pad to a multiple of the patch, row-major
patches, skip batches that are entirely nodata (all == -1), ``G(x, 0)``, ``|x_fake - x| / 2``,
band mean, ``+=`` into a float64 raster, then ``binarize`` (numpy ``>`` in float64).
``predict_scene`` keeps everything on the device: the uint16 scene is uploaded once, a kernel cuts
and normalises each batch of patches (reflect padding by index arithmetic, no padded copy), G and the
difference map run as in training, a kernel adds the tiles into a float64 raster, and
``generate_scene_mask`` zeroes nodata and compares in float64 on the device.
``predict_scene_host`` is the same computation with the reference's host-side numpy tiling and
stitching (used by the tests to pin the device path bit for bit).
"""
from __future__ import annotations

import numpy as np
import torch

from .synth import normalize_u16


def scene_patches(h: int, w: int, patch: int, crop: int = 0):
    cps = patch - 2 * crop
    pad_h = (crop, (cps - h % cps) + crop)
    pad_w = (crop, (cps - w % cps) + crop)
    origins = [(r, c) for r in range(0, h, cps) for c in range(0, w, cps)]
    return origins, pad_h, pad_w, cps


def _device_scene(full_image: np.ndarray, solver, patch: int, batch_size: int):
    """-> (scene on the device, float64 raster on the device)."""
    from . import ops
    if full_image.dtype != np.uint16:
        raise TypeError("predict_scene expects a uint16 scene (data_loader.py:76)")
    h, w, _ = full_image.shape
    origins, _, _, _ = scene_patches(h, w, patch, 0)
    dev = solver.device
    scene = torch.from_numpy(np.ascontiguousarray(full_image)).to(dev)
    org = torch.tensor(origins, dtype=torch.int32, device=dev).view(-1, 2)
    raster = torch.zeros((h, w), dtype=torch.float64, device=dev)
    # don't waste time on invalid patches (evaluate.py:104: a batch whose samples are all nodata is
    # skipped): the flags of every tile come from one launch and one copy per scene, taken while the
    # queue is still empty, instead of a device-to-host sync in front of every batch
    tile_valid = ops.scene_tile_valid(scene, org, patch).cpu().numpy()
    for s in range(0, len(origins), batch_size):
        if not tile_valid[s:s + batch_size].any():
            continue
        o = org[s:s + batch_size].contiguous()
        x, _ = ops.scene_tiles(scene, o, patch)
        ops.scene_stitch(solver.compute_difference_map(x), o, raster)
    return scene, raster


@torch.no_grad()
def predict_scene(full_image: np.ndarray, solver, patch_size: int = None, batch_size: int = 64) -> np.ndarray:
    """(H, W, bands) uint16 -> (H, W) float64 difference raster (``evaluate.py:84-123``).  Uses
    ``solver.G`` in its current train/eval mode, exactly like the reference."""
    _, raster = _device_scene(full_image, solver, patch_size or solver.image_size, batch_size)
    return raster.cpu().numpy()


@torch.no_grad()
def predict_scene_host(full_image: np.ndarray, solver, patch_size: int = None, batch_size: int = 64) -> np.ndarray:
    """``predict_scene`` with the reference's host-side tiling and stitching."""
    if full_image.dtype != np.uint16:
        raise TypeError("predict_scene expects a uint16 scene (data_loader.py:76)")
    patch = patch_size or solver.image_size
    h, w, _ = full_image.shape
    origins, pad_h, pad_w, cps = scene_patches(h, w, patch, 0)
    padded = np.pad(full_image, (pad_h, pad_w, (0, 0)), "reflect")
    difference = np.zeros((h, w))
    dev = solver.device
    for s in range(0, len(origins), batch_size):
        chunk = origins[s:s + batch_size]
        tiles = np.stack([padded[r:r + patch, c:c + patch] for r, c in chunk])
        x = torch.from_numpy(normalize_u16(tiles)).permute(0, 3, 1, 2).contiguous()
        if bool((x == -1).all()):      # don't waste time on invalid patches (evaluate.py:104)
            continue
        delta = solver.compute_difference_map(x.to(dev)).cpu().numpy()
        for i, (r, c) in enumerate(chunk):
            hh, ww = min(h - r, cps), min(w - c, cps)
            difference[r:r + hh, c:c + ww] += delta[i, :hh, :ww]
    return difference


def generate_scene_mask(full_image: np.ndarray, solver, threshold: float = 0.02, batch_size: int = 64,
                        tif_bytes: bool = False, valid_mask: np.ndarray = None):
    """difference raster -> nodata zeroed -> ``solver.binarize`` (``evaluate.py:49-51``).

    ``tif_bytes``: also return what ``write_tifs`` stores (``evaluate.py:384-390``):
    ``(difference * 255).astype(uint8)`` and the prediction as 128 / 255 with invalid pixels 0, where
    valid is ``valid_mask`` ((H, W) bool, the reference passes ``target > 0``) or, without one,
    "any band non-zero"."""
    from . import ops
    with torch.no_grad():
        scene, raster = _device_scene(full_image, solver, solver.image_size, batch_size)
        mask = ops.scene_binarize(raster, scene, threshold)
        if tif_bytes:
            vm = None if valid_mask is None else torch.from_numpy(np.ascontiguousarray(valid_mask).astype(np.uint8)).to(mask.device)
            diff_u8, mask_u8 = ops.scene_tif_bytes(raster, mask, vm, scene)
            return raster.cpu().numpy(), mask.cpu().numpy(), diff_u8.cpu().numpy(), mask_u8.cpu().numpy()
    return raster.cpu().numpy(), mask.cpu().numpy()
