"""``RandomBrightnessContrast`` (reference augmentations/brightnesscontrast.py:8-27) on device batches.  As in the
reference, ``alpha`` (drawn from ``contrast_limit``) is the BRIGHTNESS factor and ``beta`` (from ``brightness_limit``) the
CONTRAST factor; masks and key points pass through; un-apply leaves the image alone."""
from __future__ import annotations

import random

import torch

from contour_uncertainty.augmentations.augmentation import COL, Augmentation, to_tuple


class RandomBrightnessContrast(Augmentation):
    order = 2

    def __init__(self, brightness_limit=0, contrast_limit=0):
        super().__init__()
        self.brightness_limit = to_tuple(brightness_limit)
        self.contrast_limit = to_tuple(contrast_limit)

    def get_params(self, n: int = 1):
        alpha, beta = [], []
        for _ in range(n):
            alpha.append(1.0 + random.uniform(self.contrast_limit[0], self.contrast_limit[1]))
            beta.append(1.0 + random.uniform(self.brightness_limit[0], self.brightness_limit[1]))
        return {"alpha": torch.tensor(alpha, dtype=torch.float32), "beta": torch.tensor(beta, dtype=torch.float32)}

    def fill(self, table, params, sign=1.0):
        if sign > 0:
            table[:, COL["alpha"]] = params["alpha"].to(table)
            table[:, COL["beta"]] = params["beta"].to(table)
