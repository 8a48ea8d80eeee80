"""Per-landmark uncertainty projected on the contour normal -- batched over the K landmarks.

Boundary: ``projected_uncertainty`` / ``projected_uncertainty_value`` keep the signatures and return values of the
reference's ``contour_uncertainty/utils/uncertainty_projection.py:11-129``.  The computation is organised differently:
the reference walks the landmarks one by one and, for each, rotates the 2x2 covariance (and the skewness vector) into
the normal's frame with ``distributions/utils.py`` helpers before reading one matrix entry.  Rotating by the normal's
angle and reading entry (0, 0) IS the quadratic form ``n^T Sigma n``, so here all K landmarks are handled at once by
``normal_frame_moments`` (three einsums) and the spline derivative lookup is one broadcast ``argmin``.

What is computed, for a contour ``mu`` (K, 2) with covariances ``cov`` (K, 2, 2) and optional skewness ``alpha`` (K, 2):
  * ``d_k``: derivative of the interpolating cubic B-spline through the landmarks (FITPACK ``splprep(k=3, s=0)``,
    chord-length parameter ``u_k``), taken at the sample of ``linspace(0, 1.01, 1000)`` closest to ``u_k``;
  * normal ``n_k = (d_y, -d_x) / |d|``; tangent-side axis ``t_k = (-n_y, n_x)``;
  * projected variance ``n^T Sigma n``; with ``alpha``: the skewness of the 1-D marginal along ``n`` of a skew-normal
    whose shape vector is ``(alpha_x, -alpha_y)`` (image rows grow downwards):
    ``(a_n + rho a_t) / sqrt(1 + a_t^2 (1 - rho^2))`` with ``rho`` the correlation of (n, t) under Sigma
    (Azzalini's marginalisation, reference distributions/bivariateskewnormal.py:92-135);
  * ``linear_close``: the two end points RETURN the bisector of their neighbour edge and the closing chord as direction,
    while their variance / skewness still use the spline normal (the reference computes the angle first, :52-69).
"""
from __future__ import annotations

import numpy as np
from scipy import interpolate

_SPLINE_SAMPLES = np.linspace(0, 1.01, 1000)


def contour_normals(mu: np.ndarray) -> np.ndarray:
    """(K, 2) landmarks -> (K, 2) unit normals of the interpolating spline at the landmarks' parameters."""
    tck, u = interpolate.splprep([mu[:, 0], mu[:, 1]], k=3, s=0)
    nearest = np.abs(np.asarray(u)[:, None] - _SPLINE_SAMPLES[None, :]).argmin(axis=1)
    dx, dy = interpolate.splev(_SPLINE_SAMPLES[nearest], tck, der=1)
    d = np.stack([dx, dy], axis=1)
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    return np.stack([d[:, 1], -d[:, 0]], axis=1)


def normal_frame_moments(cov: np.ndarray, normals: np.ndarray, alpha=None):
    """Variance (K,) of every landmark's distribution along its normal and, with ``alpha``, the skewness (K,) of that
    1-D marginal.  float32 inputs are promoted: the result agrees with the reference's float32 rotations to ~1e-6."""
    cov = np.asarray(cov, dtype=np.float64)
    n = np.asarray(normals, dtype=np.float64)
    t = np.stack([-n[:, 1], n[:, 0]], axis=1)
    var_n = np.einsum("ki,kij,kj->k", n, cov, n)
    if alpha is None:
        return var_n, None
    var_t = np.einsum("ki,kij,kj->k", t, cov, t)
    cov_nt = np.einsum("ki,kij,kj->k", n, cov, t)
    rho = cov_nt / np.sqrt(var_n * var_t)
    a = np.asarray(alpha, dtype=np.float64) * np.array([1.0, -1.0])
    a_n, a_t = np.einsum("ki,ki->k", n, a), np.einsum("ki,ki->k", t, a)
    return var_n, (a_n + rho * a_t) / np.sqrt(1.0 + a_t * a_t * (1.0 - rho * rho))


def _closing_bisectors(mu: np.ndarray):
    """directions of the first / last landmark when the contour is closed by a straight chord"""
    def unit(x):
        return x / np.linalg.norm(x)
    chord = unit(mu[-1] - mu[0])
    return unit((unit(mu[1] - mu[0]) + chord) / 2), unit((unit(mu[-1] - mu[-2]) + chord) / 2)


def projected_uncertainty(mu, cov, alpha=None, use_eigenvalue: bool = True, all=False, linear_close=False):  # noqa: A002
    """mu (K, 2), cov (K, 2, 2)[, alpha (K, 2)] -> (projected std (K,), directions (K, 2)[, projected skewness]).

    Without ``all`` the two basal points and the apex (indices 0, K // 2, K - 1) report the sum of the square roots of
    their covariance's eigenvalues instead of a projection, and contribute no skewness entry."""
    mu = np.asarray(mu)
    k = mu.shape[0]
    normals = contour_normals(mu)
    var_n, skew_n = normal_frame_moments(cov, normals, alpha)
    std = np.sqrt(var_n)
    directions = normals.copy()
    if linear_close:
        directions[0], directions[-1] = _closing_bisectors(mu)
    projected = np.ones(k, dtype=bool)
    if not all:
        ends = np.array([0, k // 2, k - 1])
        projected[ends] = False
        std[ends] = np.sqrt(np.linalg.eigvals(np.asarray(cov, dtype=np.float64)[ends]).real).sum(axis=1)
        directions[ends] = normals[ends]          # the reference only swaps in the bisector on the projected branch
    if alpha is not None:
        return std, directions, skew_n[projected]
    return std, directions


def projected_uncertainty_value(mu, cov, use_eigenvalue: bool = True):
    """Scalar summary of a contour: the sum of its landmarks' projected standard deviations."""
    return float(np.sum(projected_uncertainty(mu, cov, use_eigenvalue=use_eigenvalue)[0]))
