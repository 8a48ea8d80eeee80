"""Seeded random-init weights with the reference's distributions, drawn in ``state_dict`` order from ONE generator.

``bench.py`` (its ``parity`` field) and the runner's synthetic configuration need weights that a committed golden vector
was computed with: the reference initialises conv / transposed-conv weights kaiming-normal(a = negative_slope), biases 0
and InstanceNorm affine (1, 0) (reference contour_uncertainty/models/nnUnet/unet2.py:309-314) and leaves ``ConfidenceNet``
at PyTorch's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (unet2.py:14-34).  Drawing tensor by tensor in ``state_dict``
order from a ``torch.Generator`` makes the result a function of the seed alone (``tests/test_seeded_init.py`` pins it to
the generator the golden vectors of ``tests/golden/train_step.npz`` were made with)."""
from __future__ import annotations

import math
from typing import Dict

import torch


def seeded_unet_state(module: torch.nn.Module, generator: torch.Generator, negative_slope: float = 1e-2) -> Dict[str, torch.Tensor]:
    gain = math.sqrt(2.0 / (1.0 + negative_slope ** 2))
    sd: Dict[str, torch.Tensor] = {}
    for name, t in module.state_dict().items():
        shape = tuple(t.shape)
        if name.endswith("norm.weight"):
            sd[name] = torch.ones(shape)
        elif name.endswith("bias"):
            sd[name] = torch.zeros(shape)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = torch.randn(shape, generator=generator) * (gain / math.sqrt(fan_in))
    return sd


def seeded_confidence_state(module: torch.nn.Module, generator: torch.Generator) -> Dict[str, torch.Tensor]:
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        wshape = shapes[name.replace("bias", "weight")]
        bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
        sd[name] = (torch.rand(shape, generator=generator) * 2 - 1) * bound
    return sd
