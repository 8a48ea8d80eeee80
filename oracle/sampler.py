"""Oracle (test infrastructure): the Gaussian posterior-shape-model contour sampler, restated from the reference's text.

  * ``pca``                    reference contour_uncertainty/sampler/posterior_shape_model/posteriorshapemodel.py:9-46
  * ``posterior_shape_model``  reference .../posteriorshapemodel.py:49-81 (row masks, sigma2 slack)
  * ``index_to_flat``          reference .../posterior_shape_model/utils.py:4-25
  * ``get_points_order``       reference .../posterior_shape_model/psm.py:43-71
  * ``merge_priors``           reference .../psm.py:424-440 (ignores its ``p`` argument)
  * ``sample_endo_contour``    reference .../psm.py:199-384 (Gaussian branch, ``complete_shape=True``, no debug plots)
  * ``sample_points``          reference .../psm.py:387-421: ``MultivariateNormal(mu, cov).rsample`` = mu + chol(cov) @ eps

psm.py itself cannot be imported (it needs the missing module ``contour_uncertainty.data.ultromics``); its
deterministic building blocks are pinned by ``tests/golden/psm_math.npz`` (outputs of the importable ``pca`` /
``posterior_shape_model`` / ``get_points_order``).  To compare a sampler implementation exactly, the standard-normal
draws ``eps`` can be supplied (``eps[k]`` is used when point k is sampled).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


def index_to_flat(indices) -> List[int]:
    if isinstance(indices, int):
        return [indices * 2, indices * 2 + 1]
    out: List[int] = []
    for i in indices:
        out.extend([i * 2, i * 2 + 1])
    return out


def get_points_order(nb_points: int = 21, nb_initial_points: int = 3, levels: Optional[int] = None):
    initial = np.round(np.linspace(0, nb_points - 1, nb_initial_points)).astype(int).tolist()
    levels = levels or int(math.log(nb_points, 2))
    all_points, order = list(initial), []
    for _ in range(levels):
        lvl = []
        for j in range(len(all_points) - 1):
            if all_points[j] + 1 != all_points[j + 1]:
                p = (all_points[j] + all_points[j + 1]) / 2
                p = math.ceil(p) if p > nb_points / 2 else math.floor(p)
                lvl.append(int(p))
        if not lvl:
            break
        all_points.extend(lvl)
        all_points.sort()
        order.append(lvl)
    return initial, order


def pca(X: Tensor, mu: Optional[Tensor] = None):
    Xc = X[..., None]
    mu = mu if mu is not None else Xc.mean(axis=0)
    diff = Xc.squeeze().T - mu
    cov = torch.einsum("ij,kj", diff, diff) / Xc.shape[0]
    vals, vecs = torch.linalg.eig(cov)
    vals, vecs = vals.real.abs(), vecs.real
    idx = vals.argsort().flip(0)
    vals, vecs = vals[idx], vecs[:, idx]
    return mu, torch.mm(vecs, torch.diag(torch.sqrt(vals)))


def posterior_shape_model(s_g: Tensor, g_indices: Sequence[int], mu: Tensor, Q: Tensor, sigma2: float = 1):
    p = len(mu)
    eye = torch.eye(p)
    mu_mask = torch.zeros(p, 1)
    mu_mask[list(g_indices)] = 1
    q_mask = torch.zeros(p, p)
    q_mask[list(g_indices)] = 1
    mu_g, Q_g, s_g = mu * mu_mask, Q * q_mask, s_g * mu_mask
    inv = torch.inverse(Q_g.T @ Q_g + sigma2 * eye)
    mu_c = mu + Q @ inv @ Q_g.T @ (s_g - mu_g)
    cov_c = sigma2 * Q @ inv @ Q.T
    return mu_c, cov_c


def merge_priors(mu1: Tensor, cov1: Tensor, mu2: Tensor, cov2: Tensor):
    w = torch.inverse(cov1 + cov2)
    sigma_f = cov1 @ w @ cov2
    mu_f = cov1 @ w @ mu2[..., None] + cov2 @ w @ mu1[..., None]
    return mu_f, sigma_f


class GaussianPSMSamplerOracle:
    def __init__(self, psm: dict, levels: int = 3, dtype=torch.float):
        """dtype=torch.float is the reference's arithmetic; float64 (with torch.set_default_dtype) gives the same
        algorithm without its rounding noise."""
        self.mean = torch.as_tensor(np.asarray(psm["scaler_mean"]), dtype=dtype)
        self.scale = torch.as_tensor(np.asarray(psm["scaler_scale"]), dtype=dtype)
        self.X_train = torch.as_tensor(np.asarray(psm["X_train"]), dtype=dtype)
        k = self.X_train.shape[1] // 2
        self.initial_points, self.points_order = get_points_order(k, levels=levels)

    def transform(self, s):
        return ((s.reshape(1, -1) - self.mean) / self.scale).reshape(s.shape)

    def inverse_transform(self, s):
        return ((s.reshape(1, -1) * self.scale) + self.mean).reshape(s.shape)

    @staticmethod
    def _draw(mu_j, cov_j, eps_j):
        return mu_j + torch.linalg.cholesky(cov_j) @ eps_j

    def sample_one(self, mu_p: Tensor, cov_p: Tensor, pca_mu: Tensor, Q: Tensor, eps: Tensor) -> Tensor:
        """One contour (K,2) given the K x 2 standard-normal draws ``eps`` (psm.py:199-384)."""
        k = mu_p.shape[0]
        contour = torch.zeros_like(mu_p)
        sampled = list(self.initial_points)
        for j in self.initial_points:
            contour[j] = self._draw(mu_p[j], cov_p[j], eps[j])
        sigmas = [1, 1, 1, 1]
        for i, points in enumerate(self.points_order):
            sampled.sort()
            if len(sampled) == k:
                break
            s_g = self.transform(contour).reshape(-1, 1)
            mu_c, cov_c = posterior_shape_model(s_g, index_to_flat(sampled), pca_mu, Q, sigma2=sigmas[i])
            mu_c = self.inverse_transform(mu_c.squeeze()).reshape(mu_p.shape)
            cov_c = cov_c * self.scale
            cov_c = torch.stack([cov_c[2 * j:2 * j + 2, 2 * j:2 * j + 2] for j in range(k)])
            mu_f, cov_f = merge_priors(mu_p, cov_p, mu_c, cov_c)
            mu_f = mu_f.squeeze(-1)
            for j in points:
                contour[j] = self._draw(mu_f[j], cov_f[j], eps[j])
            sampled.extend(points)
        sampled.sort()
        if len(sampled) != k:
            s_g = self.transform(contour).reshape(-1, 1)
            mu_c, _ = posterior_shape_model(s_g, index_to_flat(sampled), pca_mu, Q, sigma2=0.001)
            mu_c = self.inverse_transform(mu_c.squeeze()).reshape(mu_p.shape)
            rest = [j for j in range(k) if j not in sampled]
            contour[rest] = mu_c[rest]
        return contour

    def __call__(self, mu: Tensor, cov: Tensor, n: int = 1, eps: Optional[Tensor] = None,
                 generator: Optional[torch.Generator] = None) -> Tensor:
        """mu (K,2), cov (K,2,2) -> (n,K,2)  (psm.py:73-93: PCA re-centred on the predicted contour)."""
        pca_mu, Q = pca(self.X_train, self.transform(mu).reshape(-1, 1))
        if eps is None:
            eps = torch.randn(n, mu.shape[0], 2, generator=generator)
        return torch.stack([self.sample_one(mu, cov, pca_mu, Q, eps[i]) for i in range(n)])
