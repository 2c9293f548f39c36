"""Synthetic FCD inputs: numpy PCG64 only, so every machine draws the
same patches.  uint16 Landsat-like patches normalised like ``data_loader.py:375``
(Normalize(mean .5, std .5, max 65535)), an all-bands nodata rectangle in every 4th sample
(maps to exactly -1), Bernoulli labels, permuted target labels, alpha for the GP."""
import numpy as np
import torch


def normalize_u16(v: np.ndarray) -> np.ndarray:
    t = v.astype(np.float32)
    t -= np.float32(32767.5)
    t *= np.reciprocal(np.float32(32767.5), dtype=np.float32)
    return t


def synth_batch(b: int, size: int = 128, bands: int = 10, seed: int = 8888):
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


def synth_scene(h: int, w: int, bands: int = 10, seed: int = 99, border: int = 16) -> np.ndarray:
    """(H, W, bands) uint16 scene with a nodata (zero) border, like a rotated Landsat footprint."""
    rng = np.random.default_rng(seed)
    s = rng.integers(1, 65536, size=(h, w, bands), dtype=np.uint16)
    s[:border] = 0
    s[-border:] = 0
    s[:, :border] = 0
    s[:, -border:] = 0
    return s
