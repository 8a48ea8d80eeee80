"""Skew-normal uncertainty map of a predicted contour (reference contour_uncertainty/utils/skew_umap.py:11-81).

Per landmark the projected 1-D skew-normal gives 100 iso-density levels on each side of its mode (host NumPy, K x 1000
samples); the resulting 200 contours are rasterised by ONE ``cu_contour_masks`` launch and reduced by
``cu_mask_weighted_entropy`` (the reference loops 200 scipy spline fits + hole fills per frame on the host)."""
from __future__ import annotations

import numpy as np
import torch
from scipy.stats import norm, skewnorm

from contour_uncertainty.utils.uncertainty_projection import projected_uncertainty


def skew_umap_contours(mu, cov, alpha, linear_close: bool = False):
    """-> projected mode (K, 2), iso-density contours (200, K, 2), weights (200,)  (reference skew_umap.py:12-58)."""
    alpha = np.array(alpha)
    u, v, alpha_proj = projected_uncertainty(mu, cov, alpha.copy(), all=True, linear_close=linear_close)
    cov_width, resolution, n = 2, 1000, 100
    projected_mode = np.zeros_like(mu)
    values = np.linspace(0, 0.95, n)
    contours = np.zeros((2 * n, len(mu), 2))
    weights = np.zeros(2 * n)
    for index in range(len(mu)):
        p1 = mu[index] + v[index] * u[index] * cov_width
        p2 = mu[index] - v[index] * u[index] * cov_width
        x = np.linspace(-3 * u[index], 3 * u[index], resolution)
        y = skewnorm.pdf(x, alpha_proj[index], 0, u[index])
        y = y / y.max()
        mode_y, am = y.max(), y.argmax()
        mode_x = x[am]
        frac = am / len(y)
        projected_mode[index] = p1 * frac + (1 - frac) * p2
        above, below = y[x > mode_x], y[x < mode_x]
        for i, val in enumerate(values):
            val = mode_y - val
            plus = (np.argmin(np.abs(above - val)) + am) / len(y)
            minus = np.argmin(np.abs(below - val)) / len(y)
            contours[n - i - 1, index] = p1 * minus + (1 - minus) * p2
            contours[n + i, index] = p1 * plus + (1 - plus) * p2
            weights[n - i - 1] = weights[n + i] = norm.pdf(i, loc=0, scale=n / 2)
    return projected_mode, contours, weights


def skew_umap(mu, cov, alpha, shape=(256, 256), close=True, linear_close=False):
    """-> (projected mode (K, 2), uncertainty map (256, 256) float64); like the reference the map is always 256 x 256."""
    from cu_hip import ops
    projected_mode, contours, weights = skew_umap_contours(mu, cov, alpha, linear_close)
    c = torch.as_tensor(contours, dtype=torch.float32).cuda()
    packed, _ = ops.contour_masks(c, 256, 256, as_bytes=False)
    w = torch.as_tensor(weights / weights.sum(), dtype=torch.float32).cuda()
    _, ent = ops.mask_weighted_entropy(packed, 1, 256, w)
    return projected_mode, ent[0].cpu().numpy().astype(np.float64)
