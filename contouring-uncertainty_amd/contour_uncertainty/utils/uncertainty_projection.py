"""Per-landmark uncertainty projected on the contour normal (reference contour_uncertainty/utils/uncertainty_projection.py
:11-129).  Host code of the predict-step post-processing: one FITPACK spline per contour, K tiny 2x2 rotations."""
from __future__ import annotations

import numpy as np
import torch
from scipy import interpolate

from contour_uncertainty.distributions.bivariatenormal import BivariateNormal
from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal


def projected_uncertainty_value(mu, cov, use_eigenvalue: bool = True):
    uncertainties, _ = projected_uncertainty(mu, cov, use_eigenvalue=use_eigenvalue)
    return np.sum(uncertainties)


def projected_uncertainty(mu, cov, alpha=None, use_eigenvalue: bool = True, all=False, linear_close=False):  # noqa: A002
    """mu (K, 2), cov (K, 2, 2)[, alpha (K, 2)] -> (projected std per point, unit normals (K, 2)[, projected skewness]).
    Without ``all`` the two basal points and the apex get the sum of the square-rooted eigenvalues instead."""
    tck, u = interpolate.splprep([mu[:, 0], mu[:, 1]], k=3, s=0)
    unew = np.linspace(0, 1.01, 1000)
    der = np.stack(interpolate.splev(unew, tck, der=1), axis=1)
    k = mu.shape[0]
    uncertainties, projections, alpha_proj = [], [], []
    for index in range(k):
        i = np.argmin(np.abs(u[index] - unew))
        v = der[i] / np.linalg.norm(der[i])
        v = np.flip(v)
        v[1] = -v[1]
        if index in [0, k // 2, k - 1] and not all:
            w, _ = np.linalg.eig(cov[index])
            uncertainties.append(np.sum(np.sqrt(w)))
        else:
            angle = np.arctan2(v[1], v[0])          # = arctan2(cross((1, 0), v), dot((1, 0), v))
            if linear_close and index in (0, k - 1):
                nb = mu[1] - mu[0] if index == 0 else mu[-1] - mu[-2]
                other = mu[-1] - mu[0]
                v = (nb / np.linalg.norm(nb) + other / np.linalg.norm(other)) / 2
                v = v / np.linalg.norm(v)
            if alpha is not None:
                _, var_v, alpha_v = BivariateSkewNormal.marginal(mu[index], cov[index], alpha[index], axis=0,
                                                                 angle=torch.tensor(angle))
                uncertainties.append(np.sqrt(var_v))
                alpha_proj.append(alpha_v)
            else:
                _, sigma = BivariateNormal.marginal(mu[index], cov[index], axis=0, angle=torch.tensor(angle))
                uncertainties.append(np.sqrt(sigma))
        projections.append(v)
    if alpha is not None:
        return np.array(uncertainties), np.array(projections), np.array(alpha_proj)
    return np.array(uncertainties), np.array(projections)
