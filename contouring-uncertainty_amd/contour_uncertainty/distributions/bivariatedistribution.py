"""``BivariateDistribution`` base (reference distributions/bivariatedistribution.py:5-91): shape arrangement + det."""
from __future__ import annotations

import torch


class BivariateDistribution:
    @classmethod
    def _arrange_shapes(cls, x, loc, cov, alpha=None):
        """x (2,)|(N,2), loc (2,)|(N,2), cov (2,2)|(N,2,2)[, alpha (2,)|(N,2)] -> batched (reference :10-46)."""
        assert x.shape[-1] == 2 and x.ndim < 3, f"x must be (2,) or (N, 2), got {tuple(x.shape)}"
        x = x if x.ndim == 2 else x.unsqueeze(0)
        assert loc.shape[-1] == 2 and loc.ndim < 3, f"loc must be (2,) or (N, 2), got {tuple(loc.shape)}"
        loc = loc if loc.ndim == 2 else loc.unsqueeze(0)
        assert cov.shape[-2:] == (2, 2) and cov.ndim in (2, 3), f"cov must be (2,2) or (N,2,2), got {tuple(cov.shape)}"
        cov = cov if cov.ndim == 3 else cov.unsqueeze(0)
        if alpha is None:
            return x, loc, cov
        assert alpha.shape[-1] == 2 and alpha.ndim < 3, f"alpha must be (2,) or (N, 2), got {tuple(alpha.shape)}"
        alpha = alpha if alpha.ndim == 2 else alpha.unsqueeze(0)
        return x, loc, cov, alpha

    @classmethod
    def logpdf(cls, x, loc, cov, *args, **kwargs):
        raise NotImplementedError

    @classmethod
    def pdf(cls, x, loc, cov, *args, **kwargs):
        return torch.exp(cls.logpdf(x, loc, cov, *args, **kwargs))

    @classmethod
    def nll(cls, y, mu, cov, *args, **kwargs):
        raise NotImplementedError

    @classmethod
    def det(cls, matrix):
        return matrix[:, 0, 0] * matrix[:, 1, 1] - matrix[:, 0, 1] * matrix[:, 1, 0]


def _frame_axis(angle, axis: int):
    """unit vector of coordinate ``axis`` (0 | 1) of the frame turned by ``angle`` -> numpy (2,)"""
    import numpy as np
    a = float(angle)
    return np.array([np.cos(a), np.sin(a)]) if axis == 0 else np.array([-np.sin(a), np.cos(a)])


def _sigma3(cov: torch.Tensor) -> torch.Tensor:
    return torch.stack([cov[..., 0, 0], cov[..., 1, 1], cov[..., 1, 0]], -1).float().contiguous()


def _device_logpdf(x, loc, cov, alpha=None):
    """Shared body of the two logpdf classmethods: one distribution over a point set, or point-wise pairs."""
    from cu_hip import ops
    shape = x.shape[:-1]
    pts = x.reshape(-1, 2).float().contiguous()
    if alpha is None:
        _, loc, cov = BivariateDistribution._arrange_shapes(pts, loc, cov)
    else:
        _, loc, cov, alpha = BivariateDistribution._arrange_shapes(pts, loc, cov, alpha)
    dev = pts.device
    loc = loc.to(dev).float().contiguous()
    s3 = _sigma3(cov.to(dev))
    al = alpha.to(dev).float().contiguous() if alpha is not None else None
    if loc.shape[0] == 1:
        out = ops.logpdf_grid(pts, loc, s3, al)[0]
    elif loc.shape[0] == pts.shape[0]:
        out = ops.logpdf_grid(pts, loc, s3, al, pairwise=True)
    else:
        raise ValueError(f"cannot broadcast {pts.shape[0]} points against {loc.shape[0]} distributions")
    return out.reshape(shape)
