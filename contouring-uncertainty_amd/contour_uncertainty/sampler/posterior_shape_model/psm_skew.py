"""``SkewPosteriorShapeModelSampler`` on the MI355X kernels (reference sampler/posterior_shape_model/psm_skew.py:162-503).

Same constructor (``psm_path``, ``levels``, ``skew_indices``) and ``__call__(mu (B,K,2), cov (B,K,2,2), alpha (B,K,2),
n) -> (B, n, K, 2)``.  Per frame one ``cu_psm_setup`` record (PCA re-centred on the prediction, gains and conditional
covariances of every level); then ONE launch draws all n contours of all frames, one workgroup per contour: anchors by
``rvs_fast``, every other skew point by ``numerical_sampling`` (inverse-CDF draw from skew-pdf x N(mu_c, cov_c) on the
256 x 256 pixel grid, evaluated on the fly).  The reference evaluates K tables of 256 x 256 per frame plus one per
point, level and sample, and calls ``torch.multinomial`` per point in a Python loop.

Differences that are deliberate:
  * points outside ``skew_indices`` call the undefined ``merge_gaussian_priors`` in the reference (psm_skew.py:329,
    an AttributeError); the product-of-Gaussians ``merge_priors`` is used;
  * for an un-batched call the reference sets ``cov = mu[None]`` (psm_skew.py:199, a typo); ``cov[None]`` is used;
  * the categorical draw is an inverse-CDF draw of a uniform (``torch.multinomial`` consumes its generator differently:
    same distribution, different stream).
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional

import torch

from contour_uncertainty.sampler.posterior_shape_model.psm import PosteriorShapeModelSampler
from cu_hip import ops


def cov_to3(cov: torch.Tensor) -> torch.Tensor:
    """(..., 2, 2) -> (..., 3) = {xx, yy, yx} (the lower triangle, which is what Cholesky-based torch code reads)."""
    return torch.stack([cov[..., 0, 0], cov[..., 1, 1], cov[..., 1, 0]], -1).contiguous()


class SkewPosteriorShapeModelSampler(PosteriorShapeModelSampler):
    def __init__(self, psm_path: Path, levels: int = 3, skew_indices: List[int] = None):
        super().__init__(psm_path, levels)
        self.skew_indices = list(range(self.nb_points)) if skew_indices is None else list(skew_indices)
        self.grid_size = 256            # psm_skew.py:181
        self._skew_bits = sum(1 << int(k) for k in self.skew_indices)

    # ------------------------------------------------------------------------------------------------ batched entry
    def sample_batch(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor = None, n: int = 1, **kw) -> torch.Tensor:
        """mu (F,K,2), cov (F,K,2,2), alpha (F,K,2) -> (F, n, K, 2) on the GPU; keyword arguments as
        ``PosteriorShapeModelSampler.sample_batch_skew`` (eps, u, seed, prior_mu, prior_cov, use_initial_pdf, ...)."""
        if alpha is None:
            alpha = torch.zeros_like(mu)
        return self.sample_batch_skew(mu, cov, alpha, n=n, skew_bits=self._skew_bits, grid_size=self.grid_size, **kw)

    # ------------------------------------------------------------------------------------------- reference surface
    def __call__(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor, n: int = 1, debug_img=None,
                 progress_bar=False) -> torch.Tensor:
        if mu.ndim == 2:
            mu, cov = mu[None], cov[None]
            alpha = alpha[None] if alpha is not None else None
        if alpha is None:
            alpha = torch.zeros_like(mu)
        return self.sample_batch(mu, cov, alpha, n=n).to(mu.device)

    def sample_one_instant(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor, n: int = 1, debug_img=None,
                           progress_bar=False, pdfs=None, use_initial_pdf: bool = False) -> torch.Tensor:
        """(K,2), (K,2,2), (K,2) -> (n, K, 2)  (psm_skew.py:210-244).  Density tables are never materialised here:
        a caller-supplied ``pdfs`` has no equivalent - use ``sample_batch(prior_mu=..., prior_cov=...)``."""
        if pdfs is not None or use_initial_pdf:
            raise NotImplementedError("tables are evaluated inside the kernel: pass the Gaussian factor as "
                                      "sample_batch(prior_mu=, prior_cov=, use_initial_pdf=True)")
        return self.sample_batch(mu[None], cov[None], alpha[None], n=n)[0].to(mu.device)
