"""Synthetic stand-in for ``CamusContourDataModule`` (reference data/camus/datamodule.py:17-85, dataset.py:100-149).

The CAMUS HDF5 files cannot travel, so ``data=synthetic`` (config/data/synthetic.yaml) feeds the same batch contract:

  train / val item   ``img`` float32 (1, S, S) in [0, 1], ``contour`` float32 (K, 2) pixel (x = column, y = row),
                     ``gt`` int64 (S, S), ``id``, ``group``, ``frame_pos``           (dataset.py:100-149)
  predict item       one whole view: ``img`` (F, 1, S, S) with F = 2 instants (ED, ES), ``contour`` (F, K, 2),
                     ``gt`` (F, S, S), ``id`` -- the predict loader has ``batch_size=None``
                     (vital/vital/data/camus/data_module.py:93-94)

``synthetic_batch`` is the input generator of bench.py: uniform-noise images and jittered half-ellipse
contours.  The dataset items add the filled contour to the image so that a few optimisation steps have something to learn.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from contour_uncertainty._compat import ContourTags, DataParameters, Tags
from contour_uncertainty.data.camus.utils import USContourToMask, USSkewUmap, USUMap


def _arcs(n: int, size: int, k: int, g: torch.Generator):
    """n open half-ellipse contours (apex up, like an LV in an apical view), pixel (x, y), jittered by size / 128 px"""
    t = torch.linspace(0.0, torch.pi, k)[None]
    c = size / 2.0
    rx = (0.16 + 0.19 * torch.rand(n, 1, generator=g)) * size
    ry = (0.16 + 0.19 * torch.rand(n, 1, generator=g)) * size
    x = c + rx * torch.cos(t) + torch.randn(n, k, generator=g) * (size / 128.0)
    y = c - ry * torch.sin(t) + 0.15 * size + torch.randn(n, k, generator=g) * (size / 128.0)
    return torch.stack([x, y], dim=-1).clamp(1.0, size - 2.0)


def synthetic_batch(n: int, size: int, k: int = 21, seed: int = 1234, device="cpu"):
    """img ~ U[0, 1) (N, 1, S, S); contour (N, K, 2) = jittered half-ellipse in pixel (x, y)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, 1, size, size, generator=g)
    return img.to(device), _arcs(n, size, k, g).to(device)


class SyntheticContours(Dataset):
    """Deterministic items (seed + index).  ``predict=True``: items are whole views of ``frames`` instants."""

    def __init__(self, n_items: int, size: int, k: int, seed: int, predict: bool = False, frames: int = 2):
        self.n, self.size, self.k, self.seed, self.predict, self.frames = n_items, size, k, seed, predict, frames

    def __len__(self):
        return self.n

    def _frame(self, g: torch.Generator):
        from contour_uncertainty.utils.contour import linear_reconstruction
        contour = _arcs(1, self.size, self.k, g)[0]
        gt = torch.from_numpy(linear_reconstruction(contour.numpy(), (self.size, self.size)).astype(np.int64))
        img = 0.55 * torch.rand(1, self.size, self.size, generator=g) + 0.45 * gt[None].float()
        return img, contour, gt

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        if not self.predict:
            img, contour, gt = self._frame(g)
            return {Tags.img: img, ContourTags.contour: contour, Tags.gt: gt, Tags.id: f"synthetic{i:04d}-2CH_0",
                    Tags.group: f"synthetic{i:04d}", "frame_pos": 0.0}
        frames = [self._frame(g) for _ in range(self.frames)]
        return {Tags.img: torch.stack([f[0] for f in frames]), ContourTags.contour: torch.stack([f[1] for f in frames]),
                Tags.gt: torch.stack([f[2] for f in frames]), Tags.id: f"synthetic{i:04d}-2CH",
                Tags.group: f"synthetic{i:04d}", "instants": {"ED": 0, "ES": self.frames - 1}}


class SyntheticContourDataModule:
    """``data=synthetic``: the slice of ``VitalDataModule`` the runner and the tasks touch."""
    contour_to_mask_fn = USContourToMask()
    umap_fn = USUMap()
    skew_umap_fn = USSkewUmap()

    def __init__(self, size: int = 256, points_per_side: int = 11, labels: Optional[Sequence[int]] = (0, 1),
                 batch_size: int = 32, num_workers: int = 0, n_train: int = 64, n_val: int = 8, n_predict: int = 4,
                 seed: int = 1234, da: bool = False, **_unused):
        nb_points = 2 * points_per_side - 1                       # reference datamodule.py:76-84 (one LV contour)
        # label names -> vital.data.camus.config.Label values, like Label.from_proto_labels does for the CAMUS module
        names = {"bg": 0, "lv": 1, "myo": 2, "atrium": 3}
        labels = [names[lb.lower()] if isinstance(lb, str) else int(getattr(lb, "value", lb)) for lb in labels]
        self.data_params = DataParameters(in_shape=(1, size, size), out_shape=(nb_points, 2), labels=labels)
        self.size, self.k, self.batch_size, self.num_workers, self.seed = size, nb_points, batch_size, num_workers, seed
        self.counts = {"train": n_train, "val": n_val, "predict": n_predict}
        self.datasets = {}
        self._dataset = self.datasets          # the name UncertaintyTask.on_fit_start reads (reference uncertainty.py:78)
        # the reference's data augmentation and test-time augmentation sets (data/camus/datamodule.py:46-55), here applied to
        # whole batches ON THE DEVICE after the transfer: the CPU workers only read and collate
        from contour_uncertainty.augmentations import (Compose, RandomBrightnessContrast, RandomGamma, RandomRotation,
                                                       RandomTranslation)
        shape = (size, size)
        self.tta_transforms = Compose([RandomRotation(3, shape), RandomBrightnessContrast(0.2, 0.2), RandomGamma((0.8, 1.2)),
                                       RandomTranslation(5, 5)])
        self.da_transforms = Compose([RandomRotation(3, shape), RandomBrightnessContrast(0.2, 0.2), RandomGamma((0.8, 1.2)),
                                      RandomTranslation(5, 5)])
        self.transforms = self.da_transforms if da else None

    def on_after_batch_transfer(self, batch, dataloader_idx: int = 0):
        """Lightning's post-transfer hook: image, label map and key points of every item transformed with the item's own
        random parameters, on the device.  TRAINING batches only, as the reference augments its training subset only
        (data/camus/datamodule.py:46-55): ``_compat.Trainer.fit`` calls the hook from its training loop; real Lightning calls
        it for validation / test / predict batches too, where ``trainer.training`` is False."""
        if self.transforms is None or not batch[Tags.img].is_cuda:
            return batch
        if not getattr(getattr(self, "trainer", None), "training", True):
            return batch
        out = self.transforms(image=batch[Tags.img], mask=batch[Tags.gt], keypoints=batch[ContourTags.contour])
        batch = dict(batch)
        batch[Tags.img], batch[Tags.gt], batch[ContourTags.contour] = out["image"], out["mask"], out["keypoints"]
        return batch

    def setup(self, stage: Optional[str] = None):
        stage = str(getattr(stage, "value", stage) or "fit")
        if stage in ("fit", "fitting"):
            self.datasets["train"] = SyntheticContours(self.counts["train"], self.size, self.k, self.seed)
        if stage in ("fit", "fitting", "validate", "validating"):
            self.datasets["val"] = SyntheticContours(self.counts["val"], self.size, self.k, self.seed + 1)
        if stage in ("predict", "predicting", "test", "testing"):
            self.datasets["predict"] = SyntheticContours(self.counts["predict"], self.size, self.k, self.seed + 2,
                                                         predict=True)

    def _loader(self, subset: str, shuffle: bool = False):
        return DataLoader(self.datasets[subset], batch_size=self.batch_size, shuffle=shuffle,
                          num_workers=self.num_workers, drop_last=False)

    def train_dataloader(self):
        return self._loader("train", shuffle=True)

    def val_dataloader(self):
        return self._loader("val")

    def predict_dataloader(self):
        return DataLoader(self.datasets["predict"], batch_size=None, num_workers=self.num_workers)
