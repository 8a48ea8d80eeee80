"""Algorithmic work of the U-Net kernel schedule: multiply-accumulates per image, by layer class.

The roofline denominator of bench.py: counted from the layer list the engine launches (reference
models/nnUnet/unet2.py:109-170 builds the same list: two 3x3 convs per stage, a 2x2 stride-2 transposed conv and two
3x3 convs per decoder level, the 1x1 output conv), not from a timer."""
from __future__ import annotations

from typing import Dict, Sequence


def unet_filters(n_stages: int, max_filters: int = 480):
    return [min(2 ** (5 + i), max_filters) for i in range(n_stages)]


def conv_macs_per_image(strides: Sequence[int], size: int, in_channels: int = 1, num_classes: int = 21) -> Dict[str, float]:
    f = unet_filters(len(strides))
    macs = {"fwd": 0.0, "first_conv": 0.0, "convt": 0.0, "out1x1": 0.0}
    res, cin, levels = size, in_channels, []
    for i, s in enumerate(strides):                       # encoder + bottleneck: conv(cin -> f) then conv(f -> f)
        res //= s
        levels.append(res)
        macs["fwd"] += res * res * 9 * (cin + f[i]) * f[i]
        if i == 0:
            macs["first_conv"] = res * res * 9 * cin * f[0]
        cin = f[i]
    res = levels[-1]
    for lvl in range(len(strides) - 2, -1, -1):           # decoder: convT(f[lvl+1] -> f[lvl]), conv(2f -> f), conv(f -> f)
        s = strides[lvl + 1]
        macs["convt"] += res * res * f[lvl + 1] * f[lvl] * s * s
        res *= s
        macs["fwd"] += res * res * 9 * (2 * f[lvl] + f[lvl]) * f[lvl]
    macs["out1x1"] = res * res * f[0] * num_classes
    macs["fwd"] += macs["convt"] + macs["out1x1"]
    return macs


def train_step_flops_per_image(strides: Sequence[int], size: int):
    """(whole step, forward only): fwd + input gradients + weight gradients, no input gradient for the first conv."""
    m = conv_macs_per_image(strides, size)
    return 2.0 * (3.0 * m["fwd"] - m["first_conv"]), 2.0 * m["fwd"]
