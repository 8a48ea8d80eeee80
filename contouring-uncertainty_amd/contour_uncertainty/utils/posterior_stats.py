"""Moment post-processing of a predict step (boundary: reference task/regression/aleatoric.py:88-108 and
aleatoric_skew.py:65-82), vectorised over frames, epistemic members, samples and landmarks.

The reference walks N x T_e x K slices through ``np.cov``; here the same quantities are three einsums:

  total covariance of the prediction  = E_t[Sigma_t]  (aleatoric)  +  Cov_t[mu_t]  (epistemic, population form /T_e)
  sample covariance of a point        = unbiased (/(n - 1)), like ``np.cov``'s default
"""
from __future__ import annotations

import numpy as np


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def scatter(points: np.ndarray, axis: int, ddof: int):
    """(..., n along ``axis``, ..., 2) -> covariance (..., 2, 2) of the points along ``axis``, divisor n - ddof."""
    points = np.moveaxis(points, axis, -2)
    centred = points - points.mean(axis=-2, keepdims=True)
    return np.einsum("...ni,...nj->...ij", centred, centred) / (points.shape[-2] - ddof)


def total_moments(mu, cov):
    """mu (N, T_e, K, 2), cov (N, T_e, K, 2, 2) -> (mean (N, K, 2), aleatoric (N, K, 2, 2), epistemic (N, K, 2, 2)):
    law of total variance over the T_e members (aleatoric.py:90-94)."""
    mu, cov = _np(mu), _np(cov)
    return mu.mean(axis=1), cov.mean(axis=1), scatter(mu, axis=1, ddof=0)


def sample_moments_per_member(samples: np.ndarray):
    """samples (N, T_e, T_a, K, 2) -> (post_mu (N, K, 2), post_cov (N, K, 2, 2)) as the Gaussian task reports them
    (aleatoric.py:96-108): per member the unbiased covariance of its T_a samples, then the same total-variance split
    over the members."""
    member_mu = samples.mean(axis=2)                                   # (N, T_e, K, 2)
    member_cov = scatter(samples, axis=2, ddof=1)                      # (N, T_e, K, 2, 2)
    return member_mu.mean(axis=1), member_cov.mean(axis=1) + scatter(member_mu, axis=1, ddof=0)


def sample_moments_pooled(samples: np.ndarray):
    """samples (N, T_e, T_a, K, 2) -> (post_mu (N, K, 2), post_cov (N, K, 2, 2)) as the skew task reports them
    (aleatoric_skew.py:77-81): members and samples pooled into one set of T_e * T_a points per landmark."""
    n, te, ta, k, _ = samples.shape
    pooled = samples.reshape(n, te * ta, k, 2)
    return pooled.mean(axis=1), scatter(pooled, axis=1, ddof=1)
