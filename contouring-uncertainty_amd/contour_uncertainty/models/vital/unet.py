"""``vital.models.segmentation.unet.UNet`` on the MI355X kernels (reference vital/vital/models/segmentation/unet.py:9-165):
the BatchNorm + ReLU + MaxPool U-Net that ``task/model=unet`` selects for the dsnt tasks (named in
BASELINE.json's north_star as "the `vital` U-Net backbone").

Same constructor, same ``forward(x) -> (N, num_classes, H, W)`` logits, same ``state_dict`` names and shapes as the
reference module (``layer1.net.0.weight`` ... ``layer11.conv.net.5.running_var``, ``layer12.bias``: checkpoints load with
``strict=True``) -- the ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` objects below only HOLD the parameters and
buffers; ``forward`` is one autograd node that runs the kernel schedule of ``cu_hip.engine_vital`` with a hand-written
backward.  Extra keyword (not in the reference): ``compute_dtype`` = "f32" (default: the reference's default width of 16
channels at full resolution only fits the f32 kernels) | "bf16" (``init_channels >= 64``).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from cu_hip import lib as _lib
from cu_hip.engine_vital import VitalUNetEngine


def _holder_double_conv(cin: int, cout: int) -> nn.Sequential:
    # indices as in the reference's _DoubleConv: 0 conv, 1 bn, 2 relu, 3 dropout, 4 conv, 5 bn, 6 relu, 7 dropout
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.Identity(), nn.Identity(),
                         nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.Identity(), nn.Identity())


class _Holder(nn.Module):
    def __init__(self, **mods):
        super().__init__()
        for k, v in mods.items():
            self.add_module(k, v)


class _VitalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "UNet", grad_mode: bool, x: Tensor, *params: Tensor):
        P = dict(zip(module._pnames, params))
        S = dict(module.named_buffers())
        need = grad_mode and any(ctx.needs_input_grad)      # needs_input_grad ignores no_grad: the caller passes the mode
        logits, ectx = module.engine.forward(P, S, x, module.training, keep=need)
        ctx.module, ctx.ectx = module, ectx
        ctx.save_for_backward(*params)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module: UNet = ctx.module
        params = ctx.saved_tensors
        P = dict(zip(module._pnames, params))
        total = sum(p.numel() for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=dlogits.device)
        G, off = {}, 0
        for n, p in zip(module._pnames, params):
            G[n] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        if not module.training:
            raise RuntimeError("vital UNet: backward through eval-mode BatchNorm is not built (train() the module)")
        with _lib.device_guard(dlogits):
            module.engine.backward(P, G, ctx.ectx, dlogits.float().contiguous())
        ctx.ectx = None
        return (None, None, None) + tuple(G[n] for n in module._pnames)


class UNet(nn.Module):
    def __init__(self, input_shape: Tuple[int, ...], output_shape: Tuple[int, ...], init_channels: int = 32,
                 use_batchnorm: bool = True, bilinear: bool = False, dropout: float = 0.0, compute_dtype: str = "f32",
                 drop_block: bool = False):
        super().__init__()
        if not use_batchnorm or bilinear or dropout or drop_block:
            raise NotImplementedError("the HIP path builds the default variant (use_batchnorm=True, bilinear=False, "
                                      "dropout=0): no dsnt config selects another")
        cin, k = int(input_shape[0]), int(output_shape[0])
        c = init_channels
        ch = [c // 2, c, 2 * c, 4 * c, 8 * c, 16 * c]
        self.layer1 = _Holder(net=_holder_double_conv(cin, ch[0]))
        for i in range(1, 6):           # _Down: net = Sequential(MaxPool2d, _DoubleConv)
            setattr(self, f"layer{i + 1}", _Holder(net=nn.Sequential(nn.Identity(), _Holder(net=_holder_double_conv(ch[i - 1], ch[i])))))
        for i, (a, b) in enumerate(zip(ch[:0:-1], ch[-2::-1])):      # _Up(in_ch, out_ch)
            setattr(self, f"layer{7 + i}", _Holder(upsample=nn.ConvTranspose2d(a, a // 2, 2, 2),
                                                   conv=_Holder(net=_holder_double_conv(a, b))))
        self.layer12 = nn.Conv2d(ch[0], k, 1)
        for m in self.modules():        # reference unet.py:54-57
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.xavier_uniform_(m.weight)
        self._pnames = [n for n, _ in self.named_parameters()]
        dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[str(compute_dtype).lower().replace("float32", "f32").replace("bfloat16", "bf16")]
        self.engine = VitalUNetEngine(cin, k, init_channels, dtype)

    def forward(self, x: Tensor) -> Tensor:  # noqa: D102
        _lib.require_gpu()
        if not x.is_cuda:
            raise _lib.ContourHipError("UNet.forward needs a device tensor: the HIP path has no CPU fallback")
        params = [p for _, p in self.named_parameters()]
        with _lib.device_guard(x):
            return _VitalFn.apply(self, torch.is_grad_enabled(), x.float(), *params)
