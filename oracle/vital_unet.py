"""CPU restatement of the ``vital`` U-Net (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import anything under oracle/).

Follows /root/reference/vital/vital/models/segmentation/unet.py:
  * ``UNet.__init__``            :17-58   channel plan c/2, c, 2c, 4c, 8c, 16c (c = init_channels), 1x1 output conv
  * ``UNet.forward``             :60-82   five poolings down, five transposed convolutions up, skip FIRST in the concat
  * ``_DoubleConv``              :85-119  (conv3x3 pad 1 + bias -> BatchNorm2d -> ReLU -> Dropout) x 2
  * ``_Down``                    :122-133 MaxPool2d(2, 2) -> _DoubleConv
  * ``_Up``                      :136-165 ConvTranspose2d(in, in/2, 2, 2) (bias) -> F.pad to the skip's size -> cat -> _DoubleConv

Pinned by tests/golden/vital_unet.npz, written by oracle/make_golden.py from the reference module itself (loaded by
file path).  Parameter tensors are not stored: both sides regenerate them with ``init_state`` from the same seed.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def channel_plan(init_channels: int = 32) -> List[int]:
    c = init_channels
    return [c // 2, c, 2 * c, 4 * c, 8 * c, 16 * c]


def param_shapes(in_channels: int, num_classes: int, init_channels: int = 32) -> Dict[str, Tuple[int, ...]]:
    """state_dict names -> shapes, in the reference module's registration order (parameters AND BatchNorm buffers)."""
    ch = channel_plan(init_channels)
    out: Dict[str, Tuple[int, ...]] = {}

    def double_conv(prefix, cin, cout):
        for i, (a, b) in zip((0, 4), ((cin, cout), (cout, cout))):
            out[f"{prefix}.{i}.weight"] = (b, a, 3, 3)
            out[f"{prefix}.{i}.bias"] = (b,)
            out[f"{prefix}.{i + 1}.weight"] = (b,)
            out[f"{prefix}.{i + 1}.bias"] = (b,)
            out[f"{prefix}.{i + 1}.running_mean"] = (b,)
            out[f"{prefix}.{i + 1}.running_var"] = (b,)
            out[f"{prefix}.{i + 1}.num_batches_tracked"] = ()

    double_conv("layer1.net", in_channels, ch[0])
    for k in range(1, 6):
        double_conv(f"layer{k + 1}.net.1.net", ch[k - 1], ch[k])
    for k, (cin, cout) in enumerate(zip(ch[:0:-1], ch[-2::-1])):
        p = f"layer{7 + k}"
        out[f"{p}.upsample.weight"] = (cin, cin // 2, 2, 2)
        out[f"{p}.upsample.bias"] = (cin // 2,)
        double_conv(f"{p}.conv.net", cin, cout)
    out["layer12.weight"] = (num_classes, ch[0], 1, 1)
    out["layer12.bias"] = (num_classes,)
    return out


def init_state(in_channels: int, num_classes: int, init_channels: int, generator: torch.Generator) -> Dict[str, Tensor]:
    """Seeded test weights: fan-in scaled normals for convs, non-trivial affine / biases / running statistics."""
    sd: Dict[str, Tensor] = {}
    for name, shape in param_shapes(in_channels, num_classes, init_channels).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.long)
        elif name.endswith("running_var"):
            sd[name] = 0.5 + torch.rand(shape, generator=generator)
        elif name.endswith("running_mean"):
            sd[name] = 0.1 * torch.randn(shape, generator=generator)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3] if "upsample" not in name else shape[0] * 4
            sd[name] = torch.randn(shape, generator=generator) * (2.0 / fan_in) ** 0.5
        elif name.split(".")[-2] in ("1", "5") and name.endswith("weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=generator)       # BatchNorm gamma
        else:
            sd[name] = 0.1 * torch.randn(shape, generator=generator)
    return sd


def forward(sd: Dict[str, Tensor], x: Tensor, training: bool = True, momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """logits (N, K, H, W).  training=True normalises with batch statistics and UPDATES the running statistics in
    ``sd`` in place (unbiased variance, like nn.BatchNorm2d)."""

    def bn(z, p):
        rm, rv = sd[f"{p}.running_mean"], sd[f"{p}.running_var"]
        return F.batch_norm(z, rm, rv, sd[f"{p}.weight"], sd[f"{p}.bias"], training, momentum, eps)

    def double_conv(z, p):
        for i in (0, 4):
            z = F.conv2d(z, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"], padding=1)
            z = F.relu(bn(z, f"{p}.{i + 1}"))
        return z

    skips = [double_conv(x, "layer1.net")]
    for k in range(2, 7):
        skips.append(double_conv(F.max_pool2d(skips[-1], 2, 2), f"layer{k}.net.1.net"))
    out = skips.pop()
    for k in range(7, 12):
        skip = skips.pop()
        out = F.conv_transpose2d(out, sd[f"layer{k}.upsample.weight"], sd[f"layer{k}.upsample.bias"], stride=2)
        dh, dw = skip.shape[2] - out.shape[2], skip.shape[3] - out.shape[3]
        out = F.pad(out, [dw // 2, dw - dw // 2, dh // 2, dh - dh // 2])
        out = double_conv(torch.cat([skip, out], 1), f"layer{k}.conv.net")
    return F.conv2d(out, sd["layer12.weight"], sd["layer12.bias"])
