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


def contour_spline(mu: np.ndarray, n: int = 1001, close: bool = False) -> np.ndarray:
    """interpolating cubic spline through the landmarks at n parameters (reference utils/contour.py:9-25: scipy's
    ``splprep(k=3, s=0)`` / ``splev``; a contour FITPACK refuses -- duplicate consecutive points, fewer than 4 -- comes
    back as the raw landmarks, like the reference's bare ``except``).  Host SciPy: the LV + MYO branch of
    ``USContourToMask`` calls it twice per item; the thousands of sampled contours go through ``reconstruction_batch``."""
    from scipy import interpolate
    mu = np.asarray(mu, dtype=float)
    try:
        tck, _ = interpolate.splprep([mu[:, 0], mu[:, 1]], k=3, s=0)
        spline = np.array(interpolate.splev(np.linspace(0, 1.0, n), tck)).transpose()
    except Exception:      # noqa: BLE001 -- the reference swallows every FITPACK error the same way
        spline = mu
    if close:
        spline = np.concatenate((spline, spline[0][None]))
    return spline


def polygon_mask(vertex_rows, vertex_cols, shape) -> np.ndarray:
    """``poly2mask`` of the reference (data/camus/utils.py:24-28) = ``skimage.draw.polygon`` + scatter: every pixel centre
    that lies inside the closed polygon OR on its boundary (vertex / edge), by skimage's point-in-polygon rule (even-odd
    crossing counts of the ray to the right and of the ray to the left; a point whose two counts disagree in parity sits on
    an edge; ``skimage/_shared/geometry.pxd``).  scikit-image is absent from this image: the rule is restated (vectorised
    over pixels x edges with torch, in float64 like skimage); returns an int array (H, W) of 0 / 1."""
    h, w = shape
    xp = torch.as_tensor(np.asarray(vertex_cols, dtype=np.float64))
    yp = torch.as_tensor(np.asarray(vertex_rows, dtype=np.float64))
    n = xp.numel()
    out = torch.zeros((h, w), dtype=torch.bool)
    if n < 3:
        return out.numpy().astype(int)
    r0, r1 = max(int(np.floor(yp.min().item())), 0), min(int(np.ceil(yp.max().item())), h - 1)
    c0, c1 = max(int(np.floor(xp.min().item())), 0), min(int(np.ceil(xp.max().item())), w - 1)
    if r1 < r0 or c1 < c0:
        return out.numpy().astype(int)
    xprev, yprev = torch.roll(xp, 1), torch.roll(yp, 1)          # edge i runs from vertex i-1 to vertex i
    cols = torch.arange(c0, c1 + 1, dtype=torch.float64)
    eps = 1e-12
    for r in range(r0, r1 + 1):                                  # one image row at a time: (columns x edges) work
        x0 = xp[None, :] - cols[:, None]; y0 = (yp - r)[None, :].expand_as(x0)
        x1 = xprev[None, :] - cols[:, None]; y1 = (yprev - r)[None, :].expand_as(x0)
        vertex = ((x0.abs() < eps) & (y0.abs() < eps)).any(1)
        denom = y1 - y0
        t = (x0 * y1 - x1 * y0) / torch.where(denom == 0, torch.ones_like(denom), denom)
        r_cross = (((y0 > 0) != (y1 > 0)) & (t > 0)).sum(1)
        l_cross = (((y0 < 0) != (y1 < 0)) & (t < 0)).sum(1)
        edge = (r_cross & 1) != (l_cross & 1)
        out[r, c0:c1 + 1] = vertex | edge | ((r_cross & 1) == 1)
    return out.numpy().astype(int)


def contour_to_mask(contour: np.ndarray, shape, labels=None, apply_argmax: bool = True,
                    reconstruction_type: str = "linear") -> np.ndarray:
    """Host-only single-structure converter (linear reconstruction), kept for callers without a datamodule."""
    m = linear_reconstruction(contour, shape)
    return m.astype(np.int64) if apply_argmax else m.astype(np.float32)[None]
