"""Host-side helpers with the reference's names (task/regression/dsnt/utils.py).  The heavy functions of that module --
``flat_softmax`` and ``dsnt`` over (N, K, H, W) maps -- are fused into ``cu_dsnt_head_fwd/bwd`` (see cu_hip.head); what is
left here is O(N*K) coordinate arithmetic on tiny tensors."""
from __future__ import annotations

import torch

from cu_hip.head import dsnt_moments as dsnt_from_logits  # noqa: F401  (logits -> pixel mu, Sigma on the HIP kernel)


def normalized_linspace(length, dtype=None, device=None):
    """reference utils.py:50-68: cell centres in (-1, 1); normalized_linspace(4) = [-0.75, -0.25, 0.25, 0.75]."""
    if isinstance(length, torch.Tensor):
        length = length.to(device, dtype)
    first = -(length - 1.0) / length
    return torch.arange(length, dtype=dtype, device=device) * (2.0 / length) + first


def euclidean_losses(actual, target):
    """reference utils.py:80-92"""
    assert actual.size() == target.size(), "input tensors must have the same size"
    return torch.norm(actual - target, p=2, dim=-1, keepdim=False)


def normalized_to_pixel_coordinates(coords, size):
    """reference utils.py:95-105"""
    if torch.is_tensor(coords):
        size = coords.new_tensor(size).flip(-1)
    return 0.5 * ((coords + 1) * size - 1)


def pixel_to_normalized_coordinates(coords, size):
    """reference utils.py:108-118"""
    if torch.is_tensor(coords):
        size = coords.new_tensor(size).flip(-1)
    return ((2 * coords + 1) / size) - 1
