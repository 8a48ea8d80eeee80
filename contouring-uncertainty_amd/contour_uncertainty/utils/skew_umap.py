"""Skew-normal uncertainty map of a predicted contour (boundary: reference contour_uncertainty/utils/skew_umap.py:11-81).

The map is the entropy of a weighted average of 200 filled contours: for every landmark the 1-D skew-normal along its
normal (projected std ``s_k``, projected skewness ``a_k``) is tabulated on 1000 abscissae over ``[-3 s_k, 3 s_k]``; the
contour of density level ``l`` (100 levels 0 .. 0.95 below the peak) passes, on either side of the mode, through the
abscissa whose normalised density is closest to ``1 - l``, mapped onto the segment ``mu_k -+ 2 s_k n_k``.

Host side here = ONE broadcast over (landmark, level, abscissa) instead of the reference's K x 100 Python iterations;
device side = ONE ``cu_contour_masks`` launch for the 200 spline fits + fills and ``cu_mask_weighted_entropy`` for the
reduction (the reference runs 200 scipy spline fits + hole fills per frame on the host)."""
from __future__ import annotations

import numpy as np
import torch
from scipy.stats import norm, skewnorm

from contour_uncertainty.utils.uncertainty_projection import projected_uncertainty

N_LEVELS, N_ABSCISSAE, SEGMENT_HALF_WIDTH = 100, 1000, 2


def skew_umap_contours(mu, cov, alpha, linear_close: bool = False):
    """-> projected mode (K, 2), iso-density contours (2 * N_LEVELS, K, 2) ordered outermost-below ... outermost-above,
    and the contours' weights (2 * N_LEVELS,) (a half-normal in the level index, scale N_LEVELS / 2)."""
    mu = np.asarray(mu)
    s, n, a = projected_uncertainty(mu, cov, np.array(alpha), all=True, linear_close=linear_close)
    R = N_ABSCISSAE
    # (K, R) density profile of every landmark, peak-normalised
    x = np.linspace(-3 * s, 3 * s, R, axis=-1)
    y = skewnorm.pdf(x, a[:, None], 0, s[:, None])
    y = y / y.max(axis=1, keepdims=True)
    peak = y.argmax(axis=1)                                            # (K,)
    # nearest abscissa to every target level, searched strictly above / strictly below the peak
    target = (1.0 - np.linspace(0, 0.95, N_LEVELS))[None, :, None]     # (1, L, 1)
    miss = np.abs(y[:, None, :] - target)                              # (K, L, R)
    idx = np.arange(R)[None, None, :]
    above = np.where(idx > peak[:, None, None], miss, np.inf).argmin(axis=2)       # (K, L) absolute indices
    below = np.where(idx < peak[:, None, None], miss, np.inf).argmin(axis=2)
    # fractions along the segment end -> start.  The reference measures the upper index inside the slice that starts one
    # past the peak and adds the peak's index back: one abscissa short of the absolute index (kept: drop-in equality).
    f_above = (above - 1) / R
    f_below = below / R
    f_mode = peak / R
    start = mu + n * (s * SEGMENT_HALF_WIDTH)[:, None]                 # fraction 1
    end = mu - n * (s * SEGMENT_HALF_WIDTH)[:, None]                   # fraction 0

    def on_segment(f):                                                 # f (K, L) -> (L, K, 2)
        f = f.T[:, :, None]
        return start[None] * f + (1.0 - f) * end[None]

    contours = np.concatenate([on_segment(f_below)[::-1], on_segment(f_above)], axis=0)
    mode = start * f_mode[:, None] + (1.0 - f_mode[:, None]) * end
    half = norm.pdf(np.arange(N_LEVELS), loc=0, scale=N_LEVELS / 2)
    return mode, contours, np.concatenate([half[::-1], half])


def skew_umap(mu, cov, alpha, shape=(256, 256), close=True, linear_close=False):
    """-> (projected mode (K, 2), uncertainty map (256, 256) float64); like the reference the map is always 256 x 256."""
    from cu_hip import ops
    projected_mode, contours, weights = skew_umap_contours(mu, cov, alpha, linear_close)
    c = torch.as_tensor(contours, dtype=torch.float32).cuda()
    packed, _ = ops.contour_masks(c, 256, 256, as_bytes=False)
    w = torch.as_tensor(weights / weights.sum(), dtype=torch.float32).cuda()
    _, ent = ops.mask_weighted_entropy(packed, 1, 256, w)
    return projected_mode, ent[0].cpu().numpy().astype(np.float64)
