"""``BivariateSkewNormal`` on the HIP kernels (reference distributions/bivariateskewnormal.py:16-191).

logpdf / pdf / nll / affine / unit_normal_logcdf / rvs_fast keep the reference's signatures.  ``mode``, ``marginal`` and
the plotting helpers are analysis utilities outside the accelerated path."""
from __future__ import annotations

import math

import torch

from contour_uncertainty.distributions.bivariatedistribution import BivariateDistribution, _device_logpdf, _sigma3


class BivariateSkewNormal(BivariateDistribution):
    log2 = torch.log(torch.tensor(2))

    @classmethod
    def logpdf(cls, x, loc, cov, alpha):
        """log 2 + normal logpdf + log(Phi(alpha^T cov^-1/2 (x - loc)) + 1e-7)   (reference :19-34)"""
        return _device_logpdf(x, loc, cov, alpha)

    @classmethod
    def unit_normal_logcdf(cls, x):
        return torch.log(0.5 * (1 + torch.erf(x / math.sqrt(2))) + 1e-7)

    @classmethod
    def nll(cls, y, mu, cov, alpha):
        """(nll, term1, term2, term3) per point, nll = t1/2 + t2/2 - t3 (reference :51-61); y, mu, alpha (M,2,1)."""
        from cu_hip import ops
        m = mu.shape[0]
        terms = torch.empty((m, 4), dtype=torch.float32, device=mu.device)
        ops.nll_fwd_bwd(mu.reshape(m, 2).float().contiguous(), _sigma3(cov), y.reshape(m, 2).float().contiguous(),
                        alpha.reshape(m, 2).float().contiguous(), need_grad=False, terms=terms)
        return terms[:, 0], terms[:, 1], terms[:, 2], terms[:, 3]

    @classmethod
    def marginal(cls, mu, cov, alpha, axis: int, angle=torch.tensor(0), *args, **kwargs):
        """(location, variance, skewness) of the marginal along ``axis`` after rotating by -angle (reference
        distributions/bivariateskewnormal.py:92-135; the y component of alpha is negated first, as there)."""
        from contour_uncertainty.distributions.utils import cov2corr, rotate_alpha, rotate_cov
        assert axis == 0 or axis == 1
        cov = rotate_cov(cov, -angle)
        alpha = torch.tensor(alpha).clone()
        alpha[1] = -alpha[1]
        alpha = rotate_alpha(alpha, -angle)
        corr, _ = cov2corr(cov)
        corr = corr.squeeze()
        other = 1 - axis
        corr_11, corr_22, corr_12 = corr[axis, axis], corr[other, other], corr[0, 1]
        alpha_1, alpha_2 = alpha[axis], alpha[other]
        corr_22_1 = corr_22 - corr_12 * corr_12 / corr_11
        alpha_1_2 = (alpha_1 + (1 / corr_11) * corr_12 * alpha_2) / torch.sqrt(1 + alpha_2 * corr_22_1 * alpha_2)
        return mu[axis], cov[axis, axis], alpha_1_2

    @classmethod
    def rvs_fast(cls, mu, cov, alpha, size=1, eps=None, seed=None):
        """Draws via the 3-D Gaussian construction (reference :159-191) -> (size, 2)."""
        from cu_hip import ops
        n = int(size[0]) if isinstance(size, (tuple, list)) else int(size)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        out = ops.skew_rvs(mu.reshape(1, 2).float().contiguous(), _sigma3(cov.reshape(1, 2, 2)),
                           alpha.reshape(1, 2).float().contiguous(), n, eps=eps, seed=seed)
        return out[0]
