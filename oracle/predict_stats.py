"""Oracle (test infrastructure): the covariance post-processing of a predict step, restated loop for loop.

  * ``aleatoric_stats``       reference contour_uncertainty/task/regression/aleatoric.py:88-108 (Gaussian task)
  * ``aleatoric_skew_stats``  reference contour_uncertainty/task/regression/aleatoric_skew.py:65-82 (skew task)

Those modules are not importable here (pytorch_lightning / vital / medpy missing), so this is a text restatement: the
statements below keep the reference's order of operations (torch means for mu / cov, NumPy ``np.cov`` per landmark).
"""
from __future__ import annotations

import numpy as np
import torch


def aleatoric_stats(mu: torch.Tensor, cov: torch.Tensor, contour_samples: np.ndarray):
    """mu (N,T_e,K,2), cov (N,T_e,K,2,2) torch; contour_samples (N,T_e,T_a,K,2) numpy -> dict of numpy arrays."""
    n = mu.shape[0]
    mu_mean = mu.mean(dim=1, keepdim=True)                                                        # :90
    cov_al = cov.mean(1)                                                                          # :91
    cov_ep = torch.mean((mu - mu_mean)[..., None] * (mu - mu_mean)[..., None].swapaxes(-1, -2), dim=1)   # :92
    post_mu = contour_samples.mean(axis=2)                                                        # :96
    k = contour_samples.shape[3]
    post_cov = np.zeros((n, contour_samples.shape[1], k, 2, 2))                                   # :97 (21 there)
    for idx in range(contour_samples.shape[0]):                                                   # :98-102
        for i in range(contour_samples.shape[1]):
            for kk in range(k):
                post_cov[idx, i, kk] = np.cov(contour_samples[idx, i, :, kk].reshape(-1, 2).T)
    post_mu_mean = post_mu.mean(axis=1, keepdims=True)                                            # :104
    post_cov_al = post_cov.mean(1)                                                                # :105
    d = (post_mu - post_mu_mean)[..., None]
    post_cov_ep = np.mean(d * d.swapaxes(-1, -2), axis=1)                                         # :106
    return {"mu": mu.mean(dim=1).numpy(), "cov_al": cov_al.numpy(), "cov_ep": cov_ep.numpy(),
            "cov": (cov_al + cov_ep).numpy(), "post_mu": post_mu.mean(axis=1), "post_cov": post_cov_ep + post_cov_al}


def aleatoric_skew_stats(mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor, contour_samples: np.ndarray):
    n = mu.shape[0]
    mu_mean = mu.mean(dim=1, keepdim=True)                                                        # :65
    cov_al = cov.mean(1)                                                                          # :67
    cov_ep = torch.mean((mu - mu_mean)[..., None] * (mu - mu_mean)[..., None].swapaxes(-1, -2), dim=1)   # :68
    post_mu = contour_samples.mean(axis=(1, 2))                                                   # :79
    k = contour_samples.shape[3]
    post_cov = np.zeros((n, k, 2, 2))                                                             # :80
    for idx in range(contour_samples.shape[0]):                                                   # :81-83
        for kk in range(k):
            post_cov[idx, kk] = np.cov(contour_samples[idx, :, :, kk].reshape(-1, 2).T)
    return {"mu": mu.mean(dim=1).numpy(), "alpha": alpha.mean(dim=1).numpy(), "cov": (cov_ep + cov_al).numpy(),
            "post_mu": post_mu, "post_cov": post_cov}
