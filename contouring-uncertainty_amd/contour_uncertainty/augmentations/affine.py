"""``RandomRotation`` / ``RandomTranslation`` (reference augmentations/affine.py:10-109) on device batches."""
from __future__ import annotations

import random

import torch

from contour_uncertainty.augmentations.augmentation import COL, Augmentation, to_tuple


class RandomRotation(Augmentation):
    geometric, order = True, 1

    def __init__(self, degrees, image_shape=(256, 256)):
        super().__init__()
        self.degrees = to_tuple(degrees)
        self.image_shape = image_shape

    def get_params(self, n: int = 1):
        # item by item, as the reference draws it (affine.py:61-63)
        return {"angle": torch.tensor([float(torch.empty(1).uniform_(float(self.degrees[0]), float(self.degrees[1])).item())
                                       for _ in range(n)], dtype=torch.float32)}

    def fill(self, table, params, sign=1.0):
        table[:, COL["angle"]] = sign * params["angle"].to(table)

    def apply_keypoints(self, keypoints, params, sign=1.0):
        """q = o + R(angle) (p - o) about o = image_shape / 2 (affine.py:43-58)"""
        ox, oy = self.image_shape[1] / 2, self.image_shape[0] / 2
        ang = torch.deg2rad(sign * params["angle"].to(keypoints)).view(-1, *([1] * (keypoints.dim() - 2)))
        c, s = torch.cos(ang), torch.sin(ang)
        ax, ay = keypoints[..., 0] - ox, keypoints[..., 1] - oy
        return torch.stack([ox + c * ax + s * ay, oy - s * ax + c * ay], dim=-1)


class RandomTranslation(Augmentation):
    geometric, order = True, 4

    def __init__(self, dx=0, dy=0):
        super().__init__()
        self.dx = to_tuple(dx)
        self.dy = to_tuple(dy)

    def get_params(self, n: int = 1):
        tx, ty = [], []
        for _ in range(n):                   # the reference's draw order: tx then ty, per item (affine.py:104-108)
            tx.append(random.randint(self.dx[0], self.dx[1]))
            ty.append(random.randint(self.dy[0], self.dy[1]))
        return {"tx": torch.tensor(tx, dtype=torch.float32), "ty": torch.tensor(ty, dtype=torch.float32)}

    def fill(self, table, params, sign=1.0):
        table[:, COL["tx"]] = sign * params["tx"].to(table)
        table[:, COL["ty"]] = sign * params["ty"].to(table)

    def apply_keypoints(self, keypoints, params, sign=1.0):
        shape = (-1,) + (1,) * (keypoints.dim() - 2)
        off = torch.stack([params["tx"], params["ty"]], dim=-1).to(keypoints).view(shape + (2,)) * sign
        return keypoints + off
