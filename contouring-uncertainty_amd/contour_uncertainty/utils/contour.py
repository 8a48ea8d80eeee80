"""Contour -> mask rasterisation (reference contour_uncertainty/utils/contour.py).

``reconstruction`` / ``reconstruction_batch`` (reference utils/contour.py:28-40: interpolating cubic spline at 1000
parameters, round, upper clip, closing line, ``binary_fill_holes``) run on the GPU: one workgroup per contour in
``cu_contour_masks`` (csrc/masks.hip), so the thousands of MC-sampled contours of a predict step become masks without
leaving the device.  ``linear_reconstruction`` (reference utils/contour.py:43-53, used by the validation Dice) is host
NumPy: ``skimage.draw.line`` between consecutive rounded points incl. the closing edge, clip, ``binary_fill_holes``;
skimage is not required, lines are drawn with an integer Bresenham walk that visits the same pixels."""
from __future__ import annotations

import numpy as np
import torch
from scipy.ndimage import binary_fill_holes


def reconstruction_batch(contours, height: int, width: int, round_landmarks: bool = False, packed: bool = False):
    """contours (M, K, 2) as (x, y) pixels, tensor or array -> uint8 cuda tensor (M, H, W) of 0/1
    (and, with ``packed``, the (M, H, 8) int32 bit-packed masks that ``cu_mask_entropy`` reduces)."""
    from cu_hip import ops
    c = torch.as_tensor(contours, dtype=torch.float32)
    if c.ndim != 3 or c.shape[-1] != 2:
        raise ValueError(f"contours must be (M, K, 2), got {tuple(c.shape)}")
    if not c.is_cuda:
        c = c.cuda()
    pk, masks = ops.contour_masks(c, int(height), int(width), round_landmarks=round_landmarks, packed=packed)
    return (masks, pk) if packed else masks


def reconstruction(points: np.ndarray, height: int, width: int) -> np.ndarray:
    """reference utils/contour.py:28-40 for one contour (K, 2); returns an int array (H, W) like the reference."""
    return reconstruction_batch(np.asarray(points)[None], height, width)[0].cpu().numpy().astype(int)


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
    """Host-only single-structure converter (linear reconstruction), kept for callers without a datamodule."""
    m = linear_reconstruction(contour, shape)
    return m.astype(np.int64) if apply_argmax else m.astype(np.float32)[None]
