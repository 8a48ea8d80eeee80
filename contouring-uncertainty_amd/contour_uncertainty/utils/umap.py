"""Gaussian uncertainty map of a predicted contour (reference contour_uncertainty/utils/umap.py:10-33): 100 copies of the
contour displaced along the landmark normals by -2..2 projected standard deviations, each drawn (spline curve + closing
line) with the weight norm.pdf(displacement), later curves overwriting earlier ones.  The 100 curves are drawn by one
``cu_contour_masks`` launch in curve mode and combined by ``cu_mask_last_value``."""
from __future__ import annotations

import numpy as np
import torch
from scipy.stats import norm

from contour_uncertainty.utils.uncertainty_projection import projected_uncertainty


def uncertainty_map(mu_p, cov_p, shape=(256, 256), close=True):
    from cu_hip import ops
    u, v = projected_uncertainty(mu_p, cov_p, all=True)
    u, v = np.array(u), np.array(v)
    steps = np.linspace(-2, 2, 100)
    contours = mu_p[None] + v[None] * u[None, :, None] * steps[:, None, None]          # (100, K, 2)
    c = torch.as_tensor(contours, dtype=torch.float32).cuda()
    packed, _ = ops.contour_masks(c, shape[0], shape[1], as_bytes=False, mode=1 if close else 2)
    values = torch.as_tensor(norm.pdf(steps, loc=0, scale=1), dtype=torch.float32).cuda()
    return ops.mask_last_value(packed, shape[1], values).cpu().numpy().astype(np.float64)
