"""Contour -> mask rasterisation used by the validation Dice (reference utils/contour.py:43-53 ``linear_reconstruction``:
``skimage.draw.line`` between consecutive rounded points incl. the closing edge, clip, ``binary_fill_holes``).
skimage is not required: lines are drawn with an integer Bresenham walk that visits the same pixels."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import binary_fill_holes


def _line(r0: int, c0: int, r1: int, c1: int):
    """Bresenham line, end points included (the pixel set of ``skimage.draw.line``)."""
    dr, dc = abs(r1 - r0), abs(c1 - c0)
    sr, sc = (1 if r1 >= r0 else -1), (1 if c1 >= c0 else -1)
    steep = dr > dc
    if steep:
        r0, c0, r1, c1, dr, dc, sr, sc = c0, r0, c1, r1, dc, dr, sc, sr
    d = 2 * dr - dc
    rr, cc = [], []
    r, c = r0, c0
    for _ in range(dc + 1):
        if steep:
            rr.append(c), cc.append(r)
        else:
            rr.append(r), cc.append(c)
        while d >= 0 and dc > 0:
            r += sr
            d -= 2 * dc
        c += sc
        d += 2 * dr
    return np.array(rr), np.array(cc)


def linear_reconstruction(contour: np.ndarray, shape) -> np.ndarray:
    h, w = shape
    pts = np.rint(np.asarray(contour)).astype(int)          # (K, 2) as (x, y)
    mask = np.zeros((h, w), dtype=bool)
    k = len(pts)
    for i in range(k):
        x0, y0 = pts[i]
        x1, y1 = pts[(i + 1) % k]
        rr, cc = _line(y0, x0, y1, x1)
        rr = np.clip(rr, 0, h - 1)
        cc = np.clip(cc, 0, w - 1)
        mask[rr, cc] = True
    return binary_fill_holes(mask)


def contour_to_mask(contour: np.ndarray, shape, labels=None, apply_argmax: bool = True,
                    reconstruction_type: str = "linear") -> np.ndarray:
    """Single-structure (LV endocardium) stand-in for the datamodule's ``contour_to_mask_fn``."""
    m = linear_reconstruction(contour, shape)
    return m.astype(np.int64) if apply_argmax else m.astype(np.float32)[None]
