"""``BivariateSkewNormal`` on the HIP kernels (reference distributions/bivariateskewnormal.py:16-191).

logpdf / pdf / nll / affine / unit_normal_logcdf / rvs_fast keep the reference's signatures.  ``mode`` / ``marginal`` are
host-side analysis helpers of the u-map post-processing; the plotting helpers are out of scope."""
from __future__ import annotations

import math

import numpy as np
import torch

from contour_uncertainty.distributions.bivariatedistribution import (BivariateDistribution, _device_logpdf, _frame_axis,
                                                                       _sigma3)


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
        """(location, variance, skewness) of the 1-D marginal along coordinate ``axis`` of the frame turned by ``angle``
        (boundary: reference distributions/bivariateskewnormal.py:92-135; alpha_y is negated first, as there -- image
        rows grow downwards).  Closed form shared with the batched u-map code, see
        ``utils.uncertainty_projection.normal_frame_moments``."""
        assert axis == 0 or axis == 1
        from contour_uncertainty.utils.uncertainty_projection import normal_frame_moments
        var, skew = normal_frame_moments(np.asarray(cov, dtype=np.float64)[None], _frame_axis(angle, axis)[None],
                                         np.asarray(alpha, dtype=np.float64)[None])
        return mu[axis], torch.tensor(var[0], dtype=torch.float32), torch.tensor(skew[0], dtype=torch.float32)

    @classmethod
    def mode(cls, mu, cov, alpha):
        """Mode estimate of the bivariate skew-normal (boundary: reference :73-82): Azzalini's univariate approximation
        ``m0*`` at the summary shape ``alpha* = sqrt(alpha^T R alpha)`` (R = correlation matrix), pushed back along
        ``R alpha``.  The reference contracts the standard-deviation VECTOR w with ``R alpha`` (``w @ corr @ alpha``: one
        scalar) instead of scaling by diag(w), so both coordinates move by the same amount; reproduced as is so that the
        two stay interchangeable (pinned by tests/golden/skew_mode.npz, generated from the reference)."""
        mu, cov, alpha = (torch.as_tensor(v, dtype=torch.float32) for v in (mu, cov, alpha))
        std = torch.sqrt(torch.diagonal(cov))
        r_alpha = (cov / torch.outer(std, std)) @ alpha
        alpha_star = torch.sqrt(alpha @ r_alpha)
        return mu + m0(alpha_star) / alpha_star * (std @ r_alpha)

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


# ---- univariate skew-normal summaries used by ``mode`` (boundary: reference :195-219) ------------------------------
def delta(alpha):
    """alpha / sqrt(1 + alpha^2)"""
    alpha = torch.as_tensor(alpha, dtype=torch.float32)
    return alpha * torch.rsqrt(1 + alpha * alpha)


def skewness(alpha):
    """third standardised moment of SN(0, 1, alpha): (4 - pi)/2 * mean^3 / (1 - mean^2)^(3/2), mean = sqrt(2/pi) delta"""
    mean = math.sqrt(2 / math.pi) * delta(alpha)
    return (4 - math.pi) / 2 * mean ** 3 / (1 - mean * mean) ** 1.5


def m0(alpha):
    """Azzalini's closed-form approximation of the mode of SN(0, 1, alpha)"""
    alpha = torch.as_tensor(alpha, dtype=torch.float32)
    mean = math.sqrt(2 / math.pi) * delta(alpha)
    spread = torch.sqrt(1 - mean * mean)
    return mean - skewness(alpha) * spread / 2 - torch.sign(alpha) / 2 * torch.exp(-2 * math.pi / torch.abs(alpha))


def univariate_mode(mu, sigma, alpha):
    return mu + sigma * m0(alpha)
