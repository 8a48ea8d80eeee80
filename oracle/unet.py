"""Oracle (test infrastructure): the dynamic nnU-Net-style U-Net and the skew head.

Restates, with plain ``torch.nn.functional`` CPU ops on a flat ``state_dict``:
  * ``UNet.__init__/forward``      reference contour_uncertainty/models/nnUnet/unet2.py:56-208
  * ``ConvLayer/ConvBlock``        reference .../nnUnet/layers.py:167-238  (conv -> InstanceNorm(affine) -> LeakyReLU)
  * ``UpsampleBlock``              reference .../nnUnet/layers.py:389-438  (ConvTranspose k=s=2, no bias -> cat(out, skip) -> ConvBlock)
  * ``OutputBlock``                reference .../nnUnet/layers.py:441-463  (1x1 conv, no bias)
  * ``ConfidenceNet``              reference .../nnUnet/unet2.py:14-34
  * weight init                    reference .../nnUnet/unet2.py:309-314 (kaiming-normal a=negative_slope, zero bias)

Parameter names and shapes are the reference's, so a reference checkpoint's
``model.*`` / ``skew_block.*`` tensors can be fed straight in.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class UNetSpec:
    """Shape of the network (reference config/task/model/unet2.yaml)."""

    in_channels: int = 1
    num_classes: int = 21
    strides: Sequence[int] = (1, 2, 2, 2, 2, 2, 2, 2)   # first conv stride of every stage
    negative_slope: float = 1e-2
    max_filters: int = 480
    eps: float = 1e-5
    filters: List[int] = field(default_factory=list)

    def __post_init__(self):
        # unet2.py:109-111
        self.filters = [min(2 ** (5 + i), self.max_filters) for i in range(len(self.strides))]

    @property
    def n_stages(self) -> int:
        return len(self.strides)


def spec_from_cfg(input_shape, output_shape, kernels, strides, negative_slope=1e-2) -> UNetSpec:
    for k in kernels:
        assert tuple(k) == (3, 3), "only 3x3 kernels are on the dsnt path"
    s = [int(st[0]) for st in strides]
    return UNetSpec(in_channels=int(input_shape[0]), num_classes=int(output_shape[0]), strides=tuple(s),
                    negative_slope=negative_slope)


# --------------------------------------------------------------------------- parameter inventory
def conv_layer_names(prefix: str) -> List[str]:
    return [f"{prefix}.conv.weight", f"{prefix}.conv.bias", f"{prefix}.norm.weight", f"{prefix}.norm.bias"]


def param_shapes(spec: UNetSpec) -> Dict[str, Tuple[int, ...]]:
    """Ordered name -> shape, in the reference's ``state_dict()`` order."""
    f = spec.filters
    out: Dict[str, Tuple[int, ...]] = {}

    def block(prefix, cin, cout):
        for li, (a, b) in enumerate(((cin, cout), (cout, cout)), start=1):
            out[f"{prefix}.conv{li}.conv.weight"] = (b, a, 3, 3)
            out[f"{prefix}.conv{li}.conv.bias"] = (b,)
            out[f"{prefix}.conv{li}.norm.weight"] = (b,)
            out[f"{prefix}.conv{li}.norm.bias"] = (b,)

    block("input_block", spec.in_channels, f[0])
    nd = spec.n_stages - 2
    for i in range(nd):
        block(f"downsamples.{i}", f[i], f[i + 1])
    block("bottleneck", f[-2], f[-1])
    up_in = f[1:][::-1]
    up_out = f[:-1][::-1]
    for i, (ci, co) in enumerate(zip(up_in, up_out)):
        out[f"upsamples.{i}.transp_conv.weight"] = (ci, co, 2, 2)
        block(f"upsamples.{i}.conv_block", 2 * co, co)
    out["output_block.conv.weight"] = (spec.num_classes, f[0], 1, 1)
    # unet2.py:262-273: heads for decoder levels 1..len(upsamples)-1 (never used by the dsnt tasks)
    for i in range(len(up_in) - 1):
        out[f"deep_supervision_heads.{i}.conv.weight"] = (spec.num_classes, f[i + 1], 1, 1)
    return out


def confidence_param_shapes(output_size: int) -> Dict[str, Tuple[int, ...]]:
    """ConfidenceNet (unet2.py:21-30): hard-codes 480 in-channels and a 2x2 bottleneck."""
    return {
        "model.0.weight": (128, 480, 3, 3), "model.0.bias": (128,),
        "model.2.weight": (128, 128, 3, 3), "model.2.bias": (128,),
        "model.4.weight": (128, 128, 3, 3), "model.4.bias": (128,),
        "model.7.weight": (output_size, 128 * 2 * 2), "model.7.bias": (output_size,),
    }


def init_unet_state(spec: UNetSpec, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """Random init with the reference's distribution (unet2.py:309-314).

    Conv / ConvTranspose weights: kaiming-normal(a=negative_slope) => N(0, gain^2/fan_in),
    gain = sqrt(2/(1+a^2)), fan_in = size(1) * kh * kw (for ConvTranspose that is out_channels*4,
    exactly what ``nn.init.kaiming_normal_`` uses).  Conv biases 0; InstanceNorm affine = (1, 0).
    """
    sd: Dict[str, Tensor] = {}
    gain = math.sqrt(2.0 / (1.0 + spec.negative_slope ** 2))
    for name, shape in param_shapes(spec).items():
        if name.endswith("norm.weight"):
            sd[name] = torch.ones(shape)
        elif name.endswith("bias"):
            sd[name] = torch.zeros(shape)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = torch.randn(shape, generator=generator) * (gain / math.sqrt(fan_in))
    return sd


def init_confidence_state(output_size: int, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """PyTorch default init of Conv2d/Linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    sd: Dict[str, Tensor] = {}
    shapes = confidence_param_shapes(output_size)
    for name, shape in shapes.items():
        wshape = shapes[name.replace("bias", "weight")]
        fan_in = 1
        for d in wshape[1:]:
            fan_in *= d
        bound = 1.0 / math.sqrt(fan_in)
        sd[name] = (torch.rand(shape, generator=generator) * 2 - 1) * bound
    return sd


# --------------------------------------------------------------------------- forward
class _LeakyGivenMask(torch.autograd.Function):
    """LeakyReLU whose BACKWARD uses a given sign pattern.  Test aid: the derivative of LeakyReLU is discontinuous at 0,
    so two correct implementations whose pre-activations differ by one ulp disagree by a factor 1/slope on that element.
    Feeding the device's own sign pattern makes gradient comparisons exact everywhere else."""

    @staticmethod
    def forward(ctx, x, mask, slope):
        ctx.save_for_backward(mask)
        ctx.slope = slope
        return torch.where(x > 0, x, x * slope)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, g * ctx.slope), None, None


class _RoundBf16(torch.autograd.Function):
    """x -> bf16 -> f32 in forward and the same rounding of the gradient in backward: what storing an activation and
    its gradient in bf16 does (used to calibrate the production mode's expected noise)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def conv_layer(sd, prefix: str, x: Tensor, stride: int, spec: UNetSpec, drop_mask: Optional[Tensor] = None,
               taps: Optional[Dict[str, Tensor]] = None, masks: Optional[Dict[str, Tensor]] = None,
               round_bf16: bool = False) -> Tensor:
    """layers.py:199-205: conv(3x3, pad 1, bias) -> [dropout2d] -> InstanceNorm2d(affine) -> LeakyReLU."""
    w = sd[f"{prefix}.conv.weight"]
    if round_bf16:
        w = _RoundBf16.apply(w)
    y = F.conv2d(x, w, sd[f"{prefix}.conv.bias"], stride=stride, padding=1)
    if drop_mask is not None:          # Dropout2d(p=.5): whole channels zeroed, survivors x2 (layers.py:154-164)
        y = y * drop_mask
    if round_bf16:
        y = _RoundBf16.apply(y)
    if taps is not None:
        taps[f"{prefix}:z"] = y
    y = F.instance_norm(y, weight=sd[f"{prefix}.norm.weight"], bias=sd[f"{prefix}.norm.bias"], eps=spec.eps)
    if masks is not None:
        y = _LeakyGivenMask.apply(y, masks[prefix], spec.negative_slope)
    else:
        y = F.leaky_relu(y, spec.negative_slope)
    if round_bf16:
        y = _RoundBf16.apply(y)
    if taps is not None:
        taps[f"{prefix}:a"] = y
    return y


def conv_block(sd, prefix: str, x: Tensor, stride: int, spec: UNetSpec, taps=None, masks=None,
               round_bf16=False, drop=None) -> Tensor:
    """layers.py:208-238: two ConvLayers, the first carries the stage stride.  ``drop``: prefix -> (N, C) Dropout2d
    multipliers (0 or 2), applied between conv and norm as layers.py:199-202 does."""
    def dm(p):
        return None if drop is None or p not in drop else drop[p][:, :, None, None]
    y = conv_layer(sd, f"{prefix}.conv1", x, stride, spec, drop_mask=dm(f"{prefix}.conv1"), taps=taps, masks=masks,
                   round_bf16=round_bf16)
    return conv_layer(sd, f"{prefix}.conv2", y, 1, spec, drop_mask=dm(f"{prefix}.conv2"), taps=taps, masks=masks,
                      round_bf16=round_bf16)


def unet_forward(sd: Dict[str, Tensor], x: Tensor, spec: UNetSpec, bottleneck_out: bool = False,
                 taps: Optional[Dict[str, Tensor]] = None, masks: Optional[Dict[str, Tensor]] = None,
                 round_bf16: bool = False, drop: Optional[Dict[str, Tensor]] = None):
    """unet2.py:177-208 (deep supervision / ssn branches are off for the dsnt tasks)."""
    out = conv_block(sd, "input_block", x, spec.strides[0], spec, taps, masks, round_bf16)
    enc = [out]
    nd = spec.n_stages - 2
    for i in range(nd):
        out = conv_block(sd, f"downsamples.{i}", out, spec.strides[i + 1], spec, taps, masks, round_bf16, drop)
        enc.append(out)
    out = conv_block(sd, "bottleneck", out, spec.strides[-1], spec, taps, masks, round_bf16, drop)
    bott = out.clone()
    if taps is not None:
        taps["bottleneck"] = bott
    up_strides = list(spec.strides[1:])[::-1]
    for i, skip in enumerate(reversed(enc)):
        s = up_strides[i]
        wt = sd[f"upsamples.{i}.transp_conv.weight"]
        if round_bf16:
            wt = _RoundBf16.apply(wt)
        out = F.conv_transpose2d(out, wt, None, stride=s)                                          # layers.py:415-417
        if round_bf16:
            out = _RoundBf16.apply(out)
        out = torch.cat((out, skip), dim=1)                                                        # layers.py:436
        out = conv_block(sd, f"upsamples.{i}.conv_block", out, 1, spec, taps, masks, round_bf16)
        if taps is not None:
            taps[f"upsamples.{i}"] = out
    out = F.conv2d(out, sd["output_block.conv.weight"], None)                                      # layers.py:456-463
    return (out, bott) if bottleneck_out else out


def confidence_forward(sd: Dict[str, Tensor], feats: Tensor) -> Tensor:
    """unet2.py:21-34: 3 x (conv3x3 pad1 + ReLU) -> flatten -> Linear(512, out)."""
    y = feats
    for i in (0, 2, 4):
        y = F.relu(F.conv2d(y, sd[f"model.{i}.weight"], sd[f"model.{i}.bias"], padding=1))
    return F.linear(y.flatten(1), sd["model.7.weight"], sd["model.7.bias"])


# --------------------------------------------------------------------------- work model
def conv_macs_per_image(spec: UNetSpec, size: int) -> Dict[str, float]:
    """MACs per image by layer class; used by bench.py for the MFMA roofline denominator."""
    f = spec.filters
    res = size
    total = 0.0
    first = 0.0
    sizes = []
    cin = spec.in_channels
    for i, s in enumerate(spec.strides):
        res //= s
        sizes.append(res)
        m = res * res * 9 * cin * f[i] + res * res * 9 * f[i] * f[i]
        if i == 0:
            first = res * res * 9 * cin * f[i]
        total += m
        cin = f[i]
    up_in = f[1:][::-1]
    up_out = f[:-1][::-1]
    r = sizes[-1]
    convt = 0.0
    for i, (ci, co) in enumerate(zip(up_in, up_out)):
        s = list(spec.strides[1:])[::-1][i]
        convt += r * r * ci * co * s * s
        r *= s
        total += r * r * 9 * (2 * co) * co + r * r * 9 * co * co
    total += convt
    out = r * r * f[0] * spec.num_classes
    total += out
    return {"fwd": total, "first_conv": first, "convt": convt, "out1x1": out}
