"""Oracle (test infrastructure): skew-normal uncertainty map of one predicted contour, restated from the reference's text.

  * ``projected_uncertainty``  reference contour_uncertainty/utils/uncertainty_projection.py:17-129 with the marginals of
                               distributions/bivariatenormal.py:69-86 and bivariateskewnormal.py:92-135 and the 2x2
                               helpers of distributions/utils.py:38-75,132-150.  That module IS importable from the
                               reference; tests/golden/umap_projection.npz (oracle/make_golden.py umap) pins this
                               restatement to its outputs.
  * ``uncertainty_map``        reference contour_uncertainty/utils/umap.py:10-33 (needs scikit-image: not importable); spline
                               through oracle/masks.py ``contour_spline`` (1001 points), line = Bresenham.
  * ``skew_umap``              reference contour_uncertainty/utils/skew_umap.py:11-81 (needs scikit-image: not importable);
                               its 200 mask reconstructions go through oracle/masks.py.
"""
from __future__ import annotations

import numpy as np
import scipy.stats
import torch
from scipy import interpolate
from scipy.stats import norm, skewnorm

from oracle import masks as M


def _rot32(theta: float) -> torch.Tensor:
    t = torch.tensor(theta)
    c, s = torch.cos(t), torch.sin(t)
    return torch.tensor([[c, -s], [s, c]]).float()


def _marginal(cov, alpha, angle):
    """axis-0 marginal after rotating by -angle: (variance, skewness | None); float32 rotation like the reference."""
    r = _rot32(-angle)
    c = r @ torch.as_tensor(cov) @ r.T
    if alpha is None:
        return c[0, 0], None
    a = torch.tensor(alpha).clone()
    a[1] = -a[1]
    a = r @ a
    std = torch.sqrt(torch.diagonal(c))
    corr = c / torch.outer(std, std)
    c22_1 = corr[1, 1] - corr[0, 1] * corr[0, 1] / corr[0, 0]
    return c[0, 0], (a[0] + (1 / corr[0, 0]) * corr[0, 1] * a[1]) / torch.sqrt(1 + a[1] * c22_1 * a[1])


def projected_uncertainty(mu, cov, alpha=None, every=False, linear_close=False):
    tck, u = interpolate.splprep([mu[:, 0], mu[:, 1]], k=3, s=0)
    unew = np.linspace(0, 1.01, 1000)
    der = np.stack(interpolate.splev(unew, tck, der=1), axis=1)
    k = len(mu)
    unc, proj, aproj = [], [], []
    for idx in range(k):
        d = der[np.argmin(np.abs(u[idx] - unew))]
        d = d / np.linalg.norm(d)
        v = np.array([d[1], -d[0]])
        if idx in (0, k // 2, k - 1) and not every:
            unc.append(np.sum(np.sqrt(np.linalg.eig(cov[idx])[0])))
        else:
            angle = np.arctan2(v[1], v[0])
            if linear_close and idx in (0, k - 1):
                a = mu[1] - mu[0] if idx == 0 else mu[-1] - mu[-2]
                b = mu[-1] - mu[0]
                v = (a / np.linalg.norm(a) + b / np.linalg.norm(b)) / 2
                v = v / np.linalg.norm(v)
            var, al = _marginal(cov[idx], None if alpha is None else alpha[idx], float(angle))
            unc.append(np.sqrt(var))
            if alpha is not None:
                aproj.append(al)
        proj.append(v)
    if alpha is not None:
        return np.array(unc), np.array(proj), np.array(aproj)
    return np.array(unc), np.array(proj)


def skew_umap_contours(mu, cov, alpha, linear_close=False):
    """-> projected mode (K, 2), the 200 iso-density contours (200, K, 2) and their weights (200,)."""
    alpha = np.array(alpha)
    u, v, ap = projected_uncertainty(mu, cov, alpha.copy(), every=True, linear_close=linear_close)
    n, width, res = 100, 2, 1000
    levels = np.linspace(0, 0.95, n)
    contours = np.zeros((2 * n, len(mu), 2))
    weights = np.zeros(2 * n)
    mode = np.zeros_like(mu)
    for k in range(len(mu)):
        p1 = mu[k] + v[k] * u[k] * width
        p2 = mu[k] - v[k] * u[k] * width
        x = np.linspace(-3 * u[k], 3 * u[k], res)
        y = skewnorm.pdf(x, ap[k], 0, u[k])
        y = y / y.max()
        am = y.argmax()
        mode[k] = p1 * (am / res) + (1 - am / res) * p2
        for i, lv in enumerate(levels):
            val = y.max() - lv
            hi = (np.argmin(np.abs(y[x > x[am]] - val)) + am) / res
            lo = np.argmin(np.abs(y[x < x[am]] - val)) / res
            contours[n - i - 1, k] = p1 * lo + (1 - lo) * p2
            contours[n + i, k] = p1 * hi + (1 - hi) * p2
            weights[n - i - 1] = weights[n + i] = norm.pdf(i, loc=0, scale=n / 2)
    return mode, contours, weights


def skew_umap(mu, cov, alpha, linear_close=False):
    mode, contours, weights = skew_umap_contours(mu, cov, alpha, linear_close)
    rec = np.array([M.reconstruction(c, 256, 256) for c in contours])
    m = np.average(rec, axis=0, weights=weights)
    return mode, scipy.stats.entropy(np.stack([m, 1 - m]), axis=0)


def uncertainty_map(mu_p, cov_p, shape=(256, 256), close=True):
    u, v = projected_uncertainty(mu_p, cov_p, every=True)
    out = np.zeros(shape)
    for i in np.linspace(-2, 2, 100):
        mu = mu_p + v * u[..., None] * i
        c = M.contour_spline(mu, close=False)
        mi = mu.astype(int)
        rr, cc = M._line(mi[-1, 1], mi[-1, 0], mi[0, 1], mi[0, 0])
        c = c.round().astype(int).clip(max=255, min=0)
        out[c[:, 1], c[:, 0]] = norm.pdf(i, loc=0, scale=1)
        if close:
            out[rr.clip(max=255, min=0), cc.clip(max=255, min=0)] = norm.pdf(i, loc=0, scale=1)
    return out
