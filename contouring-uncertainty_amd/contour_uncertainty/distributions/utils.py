"""Small 2x2 helpers of the reference's ``distributions/utils.py`` used by the projected-uncertainty post-processing
(host code; reference distributions/utils.py:38-75 ``cov2corr``, :132-150 ``rotate_cov`` / ``rotate_alpha``)."""
from __future__ import annotations

import torch


def cov2corr(cov: torch.Tensor):
    """batch of covariance matrices -> (correlation matrices, standard deviations)"""
    if cov.ndim == 2:
        cov = cov[None]
    std = torch.sqrt(torch.diagonal(cov, dim1=1, dim2=2))
    return cov / torch.bmm(std.unsqueeze(2), std.unsqueeze(1)), std


def _rot(theta) -> torch.Tensor:
    theta = torch.as_tensor(theta)
    c, s = torch.cos(theta), torch.sin(theta)
    return torch.tensor([[c, -s], [s, c]]).float()       # float32 like the reference, whatever the input precision


def rotate_cov(cov, theta):
    r = _rot(theta)
    return r @ torch.as_tensor(cov) @ r.T


def rotate_alpha(alpha, theta):
    return _rot(theta) @ torch.as_tensor(alpha)
