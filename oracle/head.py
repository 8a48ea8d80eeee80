"""Oracle (test infrastructure): DSNT head + Gaussian / skew-normal NLL, op for op as the reference.

  * ``flat_softmax``                    reference contour_uncertainty/task/regression/dsnt/utils.py:71-77
  * ``normalized_linspace``             reference .../dsnt/utils.py:50-68
  * ``dsnt``                            reference .../dsnt/utils.py:7-47   (compute_skew branch is dead; not restated)
  * ``normalized_to_pixel_coordinates`` reference .../dsnt/utils.py:95-105
  * ``euclidean_losses``                reference .../dsnt/utils.py:80-92
  * ``get_cov_matrix``                  reference contour_uncertainty/task/regression/aleatoric.py:138-144
  * ``gauss_nll``                       reference .../dsnt/dsnt_al.py:64-74
  * ``batch_matrix_pow``                reference contour_uncertainty/distributions/utils.py:100-129 (linalg.eig route)
  * ``skew_nll``                        reference contour_uncertainty/distributions/bivariateskewnormal.py:36-61
  * ``dsnt_al_shared_step`` / ``dsnt_skew_shared_step``  reference dsnt_al.py:45-74, dsnt_skew.py:61-104
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch

Tensor = torch.Tensor


def flat_softmax(inp: Tensor) -> Tensor:
    n, k = inp.shape[:2]
    return torch.softmax(inp.reshape(n * k, -1), dim=-1).view_as(inp)


def normalized_linspace(length: int, dtype=None, device=None) -> Tensor:
    first = -(length - 1.0) / length
    return torch.arange(length, dtype=dtype, device=device) * (2.0 / length) + first


def dsnt(heatmaps: Tensor):
    """Soft-argmax moments.  Square maps assumed, exactly like the reference (utils.py:9)."""
    w = heatmaps.shape[-1]
    lin = normalized_linspace(w, dtype=heatmaps.dtype, device=heatmaps.device)[None]
    X = lin.repeat(w, 1)          # X[i, j] = lin[j]  (column = x)
    Y = X.t()
    X = X[None, None]
    Y = Y[None, None]
    hm = heatmaps.flatten(-2)
    x = torch.inner(hm, X.flatten(-2))          # (N, K, 1, 1)
    y = torch.inner(hm, Y.flatten(-2))
    coords = torch.cat([x.squeeze(2), y.squeeze(2)], dim=-1)
    Xc = X - x
    Yc = Y - y
    var_x = (hm * (Xc * Xc).flatten(-2)).sum(-1)
    var_y = (hm * (Yc * Yc).flatten(-2)).sum(-1)
    covar = (hm * (Xc * Yc).flatten(-2)).sum(-1)
    var = torch.cat([var_x[..., None], var_y[..., None]], dim=-1)
    return coords, var, covar


def normalized_to_pixel_coordinates(coords: Tensor, size) -> Tensor:
    if torch.is_tensor(coords):
        size = coords.new_tensor(size).flip(-1)
    return 0.5 * ((coords + 1) * size - 1)


def euclidean_losses(actual: Tensor, target: Tensor) -> Tensor:
    return torch.norm(actual - target, p=2, dim=-1, keepdim=False)


def get_cov_matrix(var_x: Tensor, var_y: Tensor, covar_xy=0) -> Tensor:
    S = torch.zeros((var_x.shape[0], var_x.shape[1], 2, 2), device=var_x.device, dtype=var_x.dtype)
    S[:, :, 0, 0] = var_x
    S[:, :, 0, 1] = covar_xy
    S[:, :, 1, 0] = covar_xy
    S[:, :, 1, 1] = var_y
    return S


def head_moments(logits: Tensor, covar: bool = True):
    """logits (N,K,H,W) -> pixel mu (N,K,2), pixel Sigma (N,K,2,2).  dsnt_al.py:52-60."""
    image_size = logits.shape[2]
    hm = flat_softmax(logits)
    coords, var, cv = dsnt(hm)
    cv = cv if covar else 0
    mu = normalized_to_pixel_coordinates(coords, image_size)
    pvar = var * (image_size / 2) ** 2
    pcov = cv * (image_size / 2) ** 2
    sigma = get_cov_matrix(pvar[..., 0], pvar[..., 1], pcov)
    return mu, sigma


def gauss_nll(mu: Tensor, sigma: Tensor, y: Tensor, mse_weight: float = 1.0, log_penalty_weight: float = 1.0,
              literal_broadcast: bool = False) -> Dict[str, Tensor]:
    """dsnt_al.py:62-74.  mu,y (N,K,2); sigma (N,K,2,2).

    The reference adds a (NK,) tensor to a (NK,1,1) tensor, which broadcasts to (NK,1,NK); the mean of that equals
    mean(t1)+mean(t2) in exact arithmetic.  ``literal_broadcast=True`` reproduces the literal
    expression (test use only: it builds an (NK)^2 temporary).
    """
    mu_flat = torch.flatten(mu, 0, 1).unsqueeze(-1)
    y_flat = torch.flatten(y, 0, 1).unsqueeze(-1)
    S = torch.flatten(sigma, 0, 1)
    t1 = log_penalty_weight * torch.log(torch.det(S))
    d = mu_flat - y_flat
    t2 = mse_weight * ((d.transpose(-1, -2) @ torch.inverse(S)) @ d)
    loss = (t1 + t2).mean() if literal_broadcast else t1.mean() + t2.mean()
    return {"loss": loss, "loss_term1": t1.mean(), "loss_term2": t2.mean()}


def batch_matrix_pow(matrix: Tensor, p: float) -> Tensor:
    vals, vecs = torch.linalg.eig(matrix)
    vals_pow = vals.contiguous().pow(p).real
    vecs = vecs.real
    return torch.matmul(vecs, torch.matmul(torch.diag_embed(vals_pow), torch.inverse(vecs)))


def skew_affine(x: Tensor, loc: Tensor, cov: Tensor, alpha: Tensor) -> Tensor:
    return torch.bmm(alpha.transpose(-1, -2), batch_matrix_pow(cov, -0.5)) @ (x - loc)


def unit_normal_logcdf(x: Tensor) -> Tensor:
    cdf = 0.5 * (1 + torch.erf(x / math.sqrt(2)))
    return torch.log(cdf + 1e-7)


def skew_nll_terms(y: Tensor, mu: Tensor, cov: Tensor, alpha: Tensor):
    """BivariateSkewNormal.nll (bivariateskewnormal.py:51-61); y,mu,alpha (M,2,1), cov (M,2,2)."""
    t1 = torch.log(torch.det(cov)).squeeze()
    t2 = (((mu - y).transpose(-1, -2) @ torch.inverse(cov)) @ (mu - y)).squeeze()
    z = skew_affine(y, mu, cov, alpha)
    t3 = unit_normal_logcdf(z.squeeze()).squeeze()
    nll = 0.5 * t1 + 0.5 * t2 - t3
    return nll, t1, t2, t3


def scatter_alpha(a: Tensor, n_points: int, skew_indices: Optional[Sequence[int]]) -> Tensor:
    """dsnt_skew.py:68-71: (N, K*, 2) head output scattered into zeros (N, K, 2)."""
    idx = list(range(n_points)) if skew_indices is None else list(skew_indices)
    alpha = torch.zeros(a.shape[0], n_points, 2, device=a.device, dtype=a.dtype)
    alpha[:, idx, :] = a.view(a.shape[0], len(idx), 2)
    return alpha


def dsnt_al_loss(logits: Tensor, y: Tensor, covar: bool = True, mse_weight: float = 1.0,
                 log_penalty_weight: float = 1.0) -> Dict[str, Tensor]:
    """dsnt_al.py:45-74 after ``self.model(x)``."""
    mu, sigma = head_moments(logits, covar)
    logs = gauss_nll(mu, sigma, y, mse_weight, log_penalty_weight)
    logs["distance_loss"] = euclidean_losses(mu, y).mean()
    return logs


def dsnt_skew_loss(logits: Tensor, alpha: Tensor, y: Tensor, covar: bool = True) -> Dict[str, Tensor]:
    """dsnt_skew.py:73-104 after the model + skew head; alpha is the scattered (N,K,2) tensor."""
    mu, sigma = head_moments(logits, covar)
    mu_flat = torch.flatten(mu, 0, 1).unsqueeze(-1)
    y_flat = torch.flatten(y, 0, 1).unsqueeze(-1)
    cov_flat = torch.flatten(sigma, 0, 1)
    alpha_flat = torch.flatten(alpha, 0, 1).unsqueeze(-1)
    alpha_norm = torch.norm(alpha_flat, dim=-1).mean()
    nll, t1, t2, t3 = skew_nll_terms(y_flat, mu_flat, cov_flat, alpha_flat)
    return {"loss": nll.mean(), "distance_loss": euclidean_losses(mu, y).mean(), "loss_term1": t1.mean(),
            "loss_term2": t2.mean(), "loss_term3": t3.mean(), "alpha_norm": alpha_norm}
