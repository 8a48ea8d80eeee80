"""``BivariateNormal`` on the HIP kernels (reference distributions/bivariatenormal.py:12-90)."""
from __future__ import annotations

import numpy as np
import torch

from contour_uncertainty.distributions.bivariatedistribution import (BivariateDistribution, _device_logpdf, _frame_axis,
                                                                       _sigma3)


class BivariateNormal(BivariateDistribution):
    log2pi = torch.log(torch.tensor(2) * torch.pi)

    @classmethod
    def logpdf(cls, x, loc, cov, *args, **kwargs):
        """-log(2 pi) - log det(cov)/2 - d^T cov^-1 d / 2   (reference :15-32)"""
        return _device_logpdf(x, loc, cov)

    @classmethod
    def nll(cls, y, mu, cov, *args, **kwargs):
        """log det + d^T cov^-1 d per point (reference :39-43; y, mu (M,2,1), cov (M,2,2)) -> (nll, term1, term2)."""
        from cu_hip import ops
        m = mu.shape[0]
        terms = torch.empty((m, 4), dtype=torch.float32, device=mu.device)
        ops.nll_fwd_bwd(mu.reshape(m, 2).float().contiguous(), _sigma3(cov), y.reshape(m, 2).float().contiguous(), None,
                        need_grad=False, terms=terms)
        return terms[:, 0], terms[:, 1], terms[:, 2]

    @classmethod
    def mode(cls, mu, cov, *args, **kwargs):
        return mu

    @classmethod
    def marginal(cls, mu, cov, axis: int, angle=torch.tensor(0), *args, **kwargs):
        """(mean, variance) of the 1-D marginal along coordinate ``axis`` of the frame turned by ``angle`` (boundary:
        reference :69-86).  The variance along a unit vector e is the quadratic form e^T cov e; no rotated matrix is
        built (``utils.uncertainty_projection.normal_frame_moments`` does the same for K landmarks at once)."""
        assert axis == 0 or axis == 1
        from contour_uncertainty.utils.uncertainty_projection import normal_frame_moments
        var, _ = normal_frame_moments(np.asarray(cov, dtype=np.float64)[None], _frame_axis(angle, axis)[None])
        return mu[axis], torch.tensor(var[0], dtype=torch.float32)

    @classmethod
    def rvs(cls, mu, cov, size=(1,)):
        """MultivariateNormal(mu, cov).sample(size) (reference :89-90): mu + chol(cov) eps, via the skew kernel with
        alpha = 0 (which then never flips a draw's sign in distribution)."""
        from cu_hip import ops
        n = int(size[0]) if not isinstance(size, int) else size
        mu2 = mu.reshape(1, 2).float().contiguous()
        out = ops.skew_rvs(mu2, _sigma3(cov.reshape(1, 2, 2)), torch.zeros_like(mu2), n,
                           seed=int(torch.randint(0, 2 ** 62, (1,)).item()))
        return out[0]
