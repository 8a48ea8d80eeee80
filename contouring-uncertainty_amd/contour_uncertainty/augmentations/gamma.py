"""``RandomGamma`` (reference augmentations/gamma.py:10-23) on device batches: img ** gamma, clamped to [0, 1]."""
from __future__ import annotations

import random

import torch

from contour_uncertainty.augmentations.augmentation import COL, Augmentation, to_tuple


class RandomGamma(Augmentation):
    order = 3

    def __init__(self, gamma_limit=(0.99, 1.01)):
        super().__init__()
        self.gamma_limit = to_tuple(gamma_limit)

    def get_params(self, n: int = 1):
        return {"gamma": torch.tensor([random.uniform(self.gamma_limit[0], self.gamma_limit[1]) for _ in range(n)],
                                      dtype=torch.float32)}

    def fill(self, table, params, sign=1.0):
        if sign > 0:
            table[:, COL["gamma"]] = params["gamma"].to(table)
