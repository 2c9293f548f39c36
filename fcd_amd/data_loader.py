"""Input pipeline of the FCD train / validation loop with the patch set resident in HBM.

Mirrors the pieces of the reference's ``data_loader.py`` that sit directly in front of
``training_step``:

* ``L8BiomePatchPool``  <- ``L8BiomeHDFDataset`` (``data_loader.py:177-316``): same arrays
  (``images`` (N,H,W,C) uint16, ``masks`` (N,H,W) uint8 with 0 = invalid / 1 = clear /
  2 = cloudy, ``labels`` (N,) uint8 -- ``prepare_landsat8_biome.py:136-138``), same
  ``only_cloudy`` / ``keep_ratio`` / ``shuffle`` index selection (``:231-243``).  The reference
  keeps the arrays in host shared memory (``h5_mem``) and lets DataLoader workers run
  albumentations per sample; here the uint16 arrays live in device memory (a 128x128x10 patch is
  320 KiB, so 288 GB of HBM hold ~850 k patches) and one kernel launch
  (``fcd_patch_batch``) gathers, flips, shifts, normalises and transposes a whole batch.
* ``PatchLoader``       <- ``get_loader`` (``:398-428``): batch size doubled outside training,
  shuffling in training, HorizontalFlip / VerticalFlip with p = 0.5 each (``:372-373``).
  The reference's ``Affine(translate_percent=0.2)`` (``:374``) interpolates at sub-pixel shifts;
  here the shift is rounded to whole pixels (``max_shift``).  NOTE the default is 0 = no
  translation, unlike the reference, whose training transform always shifts by up to 20 % (and
  thereby feeds nodata pixels into G's invalid mask): pass ``max_shift=round(0.2 * H)`` to train
  with the reference's augmentation strength.
* ``on_before_batch_transfer`` <- ``L8BiomeDataModule.on_before_batch_transfer`` (``:56-71``).

HDF5 itself is a storage format and out of scope; ``from_hdf5`` is a thin convenience that needs
``h5py`` (absent from the build image, so it is not exercised by the tests) and ``from_arrays`` /
``from_npy`` take the same three arrays from numpy files.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, Optional

import numpy as np
import torch

from . import ops


def on_before_batch_transfer(batch: Dict[str, torch.Tensor], generator: Optional[torch.Generator] = None):
    """``{'image', 'label'[, 'mask']}`` -> ``(x_real, c_org, c_trg, label_org, label_trg[, mask])``
    (``data_loader.py:56-71``): target labels are a random permutation of the batch's labels."""
    x_real, label_org = batch["image"], batch["label"]
    if label_org.dim() == 1:
        label_org = label_org.unsqueeze(1)
    rand_idx = torch.randperm(label_org.size(0), generator=generator)
    label_trg = label_org[rand_idx.to(label_org.device)]
    c_org = label_org.clone()
    c_trg = label_trg.clone()
    if "mask" in batch:
        return x_real, c_org, c_trg, label_org, label_trg, batch["mask"]
    return x_real, c_org, c_trg, label_org, label_trg


class L8BiomePatchPool:
    """uint16 patch set in device memory; ``gather`` is ``__getitem__`` + transform + collate."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, masks: Optional[np.ndarray] = None, device="cuda",
                 only_cloudy: bool = False, keep_ratio: float = 1.0, shuffle: bool = False, seed: int = 42):
        if images.dtype != np.uint16 or images.ndim != 4:
            raise TypeError("images must be (N, H, W, C) uint16")
        if labels.shape != (images.shape[0],):
            raise ValueError("labels must be (N,)")
        if masks is not None and (masks.dtype != np.uint8 or masks.shape != images.shape[:3]):
            raise TypeError("masks must be (N, H, W) uint8")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("L8BiomePatchPool keeps the patches in GPU memory; there is no CPU path")
        labels = np.asarray(labels)
        # data_loader.py:231-243
        if only_cloudy:
            indices = np.where(labels[:] == 1)[0]
        else:
            indices = np.arange(labels.shape[0])
        if keep_ratio < 1.0:
            np.random.default_rng(seed).shuffle(indices)
            indices = indices[: int(keep_ratio * len(indices))]
        elif shuffle:
            np.random.default_rng(seed).shuffle(indices)
        self.indices = torch.from_numpy(indices.astype(np.int64)).to(device)
        self.images = torch.from_numpy(np.ascontiguousarray(images)).to(device)
        self.masks = None if masks is None else torch.from_numpy(np.ascontiguousarray(masks)).to(device)
        self.labels = torch.from_numpy(labels.astype(np.float32)).to(device)     # torch.as_tensor(label).float(), :297
        self.device = device

    @classmethod
    def from_npy(cls, root: str, mode: str = "train", **kw) -> "L8BiomePatchPool":
        """``{root}/{mode}_images.npy``, ``_labels.npy`` and (optional) ``_masks.npy``, memory-mapped."""
        images = np.load(os.path.join(root, f"{mode}_images.npy"), mmap_mode="r")
        labels = np.load(os.path.join(root, f"{mode}_labels.npy"))
        mpath = os.path.join(root, f"{mode}_masks.npy")
        masks = np.load(mpath, mmap_mode="r") if os.path.exists(mpath) else None
        return cls(images, labels, masks, **kw)

    @classmethod
    def from_hdf5(cls, path: str, mode: str = "train", **kw) -> "L8BiomePatchPool":
        """``l8biome.h5`` as written by ``prepare_landsat8_biome.py:129-140`` (needs h5py)."""
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError("from_hdf5 needs h5py; convert the three datasets to .npy and use from_npy") from e
        with h5py.File(path, "r") as f:
            g = f[mode]
            return cls(g["images"][...], g["labels"][...], g["masks"][...], **kw)

    def __len__(self) -> int:
        return int(self.indices.shape[0])

    _bad_positions = None

    def check_positions(self) -> None:
        """Raise IndexError if any device-resident ``positions`` handed to :meth:`gather` since the last call was out of
        range (one device-to-host copy; host-resident positions are checked in ``gather`` itself)."""
        bad, self._bad_positions = self._bad_positions, None
        if bad is not None and bool(bad.item()):
            raise IndexError("patch position outside the pool (device-resident positions; the gather used clamped ones)")

    def gather(self, positions: torch.Tensor, aug: Optional[torch.Tensor] = None, ret_mask: bool = False):
        """positions: int64 positions into the pool's (selected, possibly shuffled) index list."""
        if positions.numel() == 0:
            raise ValueError("empty batch")
        if not positions.is_cuda:
            # range check on the host, before the upload: no device-to-host sync in the batch loop
            if int(positions.min()) < 0 or int(positions.max()) >= len(self):
                raise IndexError("patch position outside the pool")
            positions = positions.to(self.device, torch.int64)
        else:
            # positions already on the device: no host sync here, and no device-side assert either (it would abort
            # the HIP context and poison the process): out-of-range positions are clamped for the gather and
            # remembered in a device flag that check_positions() turns into the IndexError
            positions = positions.to(self.device, torch.int64)
            bad = ((positions < 0) | (positions >= len(self))).any()
            self._bad_positions = bad if self._bad_positions is None else (self._bad_positions | bad)
            positions = positions.clamp(0, len(self) - 1)
        index = self.indices[positions].contiguous()
        if aug is not None:
            aug = aug.to(self.device, torch.int32).contiguous()
        if ret_mask and self.masks is None:
            raise RuntimeError("this pool holds no masks")
        x, m = ops.patch_batch(self.images, index, aug, self.masks if ret_mask else None)
        out = {"image": x, "label": self.labels[index]}
        if ret_mask:
            out["mask"] = m
        return out


class PatchLoader:
    """Batches of a pool in DataLoader order (``get_loader``, ``data_loader.py:398-428``)."""

    def __init__(self, pool: L8BiomePatchPool, batch_size: int = 16, mode: str = "train", shuffle: Optional[bool] = None,
                 force_no_aug: bool = False, ret_mask: bool = False, max_shift: int = 0,
                 generator: Optional[torch.Generator] = None, rank: int = 0, world_size: int = 1, seed: int = 0):
        if mode != "train":
            batch_size *= 2                              # :404-405
        self.pool, self.batch_size, self.mode = pool, batch_size, mode
        self.shuffle = (mode == "train") if shuffle is None else shuffle
        self.augment = mode == "train" and not force_no_aug   # :371
        self.ret_mask, self.max_shift = ret_mask, int(max_shift)
        self.generator = generator
        self.rank, self.world_size = rank, world_size    # DistributedSampler-style striding
        self.seed, self.epoch = int(seed), 0             # the shuffle every rank must agree on

    def __len__(self) -> int:
        n = (len(self.pool) + self.world_size - 1) // self.world_size
        return (n + self.batch_size - 1) // self.batch_size

    def draw_aug(self, b: int) -> torch.Tensor:
        """(b, 3) int32 {flags, dy, dx}: each flip with p = 0.5, shift uniform in [-max_shift, max_shift]."""
        g = self.generator
        flags = (torch.rand(b, generator=g) < 0.5).int() + 2 * (torch.rand(b, generator=g) < 0.5).int()
        if self.max_shift > 0:
            shift = torch.randint(-self.max_shift, self.max_shift + 1, (b, 2), generator=g, dtype=torch.int32)
        else:
            shift = torch.zeros((b, 2), dtype=torch.int32)
        return torch.cat([flags.view(b, 1), shift], dim=1)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        n = len(self.pool)
        if not self.shuffle:
            order = torch.arange(n)
        elif self.world_size > 1:
            # every rank draws the SAME permutation (seed + epoch, as DistributedSampler.set_epoch does),
            # so the rank-strided slices partition it; `generator` only drives the augmentation draws
            order = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch))
        else:
            order = torch.randperm(n, generator=self.generator)
        self.epoch += 1
        if self.world_size > 1:
            pad = (-n) % self.world_size                # wrap around like DistributedSampler
            order = torch.cat([order, order[:pad]])[self.rank::self.world_size]
        for s in range(0, order.numel(), self.batch_size):
            pos = order[s:s + self.batch_size]
            aug = self.draw_aug(pos.numel()) if self.augment else None
            yield self.pool.gather(pos, aug, self.ret_mask)
