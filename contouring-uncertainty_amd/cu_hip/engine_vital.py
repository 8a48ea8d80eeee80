"""Kernel schedule of the ``vital`` U-Net (reference vital/vital/models/segmentation/unet.py:9-165) on the MI355X kernels:
the secondary backbone of the dsnt tasks (``task/model=unet``; BatchNorm + ReLU + MaxPool, 7.8 M parameters).

Every operator is a launch of the same C ABI the nnU-Net path uses:

  * 3x3 convolutions (+ bias), the concat of the skip connection (two source pointers, skip FIRST), their input and weight
    gradients: ``cu_conv_gemm`` / ``cu_conv_wgrad`` (first layer, Cin = 1: ``cu_conv_c1_*``);
  * ``nn.BatchNorm2d`` + ``nn.ReLU``: the InstanceNorm kernels on the tensor viewed as ONE image of N*H*W pixels -- batch
    statistics are per-channel statistics of that image, the backward formula is the same; LeakyReLU slope 0.  Running
    statistics (momentum 0.1, unbiased variance) are updated from the kernel's mean / rstd; eval mode builds the
    per-channel scale / shift from them;
  * ``nn.MaxPool2d(2, 2)``: ``cu_maxpool2_fwd / _bwd``;
  * ``nn.ConvTranspose2d(C, C/2, 2, 2)`` (+ bias): the parity forms of ``cu_conv_gemm`` (one pass when C/2 is a multiple
    of 32, four parity launches otherwise) and the 4-tap weight gradient;
  * the 1x1 output convolution (+ bias) writes NCHW float32 logits.

Channel counts must be multiples of 32 in bf16 and of 16 in f32 (the kernels' K-chunk).  The reference default
(``init_channels=32``: 16 channels at full resolution) therefore runs natively in f32; in bf16 the 16-channel tensors are
carried as 32 channels whose upper half is identically zero: ``_pad_params`` builds zero-padded copies of the eleven
parameter tensors that touch them (padded BatchNorm channels get gamma 1 / beta 0, so they stay zero through the ReLU),
and ``backward`` slices the gradients back.  ``dropout > 0`` and ``bilinear=True`` are refused (no dsnt config sets them).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .engine import TAPS3, TAPS3_D, TAPS3_W
from .ops import Act

Tensor = torch.Tensor
CONVT_TAPS = [(dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)]


@dataclass
class _Layer:
    prefix: str                 # "<...>.net": conv at .<i>, BatchNorm at .<i + 1>
    i: int
    srcs: List[Act]
    out: Act                    # z viewed (1, N*H, W, C) + batch statistics + materialised ReLU output
    shape: Tuple[int, int, int, int]


@dataclass
class VitalCtx:
    img: Tensor
    layers: List[_Layer] = field(default_factory=list)
    pools: List[Tensor] = field(default_factory=list)            # argmax bytes of the five poolings
    ups: List[Tuple[str, Act, Act]] = field(default_factory=list)  # (prefix, source, upsampled)
    last: Optional[Act] = None
    P: Optional[Dict[str, Tensor]] = None


class VitalUNetEngine:
    def __init__(self, in_channels: int, num_classes: int, init_channels: int, dtype: torch.dtype, eps: float = 1e-5,
                 momentum: float = 0.1):
        if in_channels != 1:
            raise NotImplementedError("the HIP path serves single-channel echo images (Cin = 1)")
        c = init_channels
        self.ch = [c // 2, c, 2 * c, 4 * c, 8 * c, 16 * c]
        self.num_classes, self.dtype, self.eps, self.momentum = num_classes, dtype, eps, momentum
        self.pad16 = dtype == torch.bfloat16 and self.ch[0] == 16       # carry the 16-channel tensors as 32 channels
        need = 32 if dtype == torch.bfloat16 else 16
        if self.ch[0] % need and not self.pad16:
            raise NotImplementedError(f"{self.ch[0]} channels at full resolution: the {dtype} kernels need multiples of {need}")
        if num_classes > 32:
            raise NotImplementedError("at most 32 output maps")
        self._opcache: Dict[str, Tuple] = {}

    # ------------------------------------------------------------------------------------------------ 16 -> 32 channel padding
    # name -> (dims padded on the output side, dims padded on the input side): conv weights are (O, I, kh, kw), transposed
    # conv weights (I, O, kh, kw); "cat" = the input is the concat [skip 16 | up 16] -> [skip 16, 0 x 16, up 16, 0 x 16]
    _PAD = {"layer1.net.0.weight": ("o", None), "layer1.net.0.bias": ("v", None), "layer1.net.1.weight": ("g", None),
            "layer1.net.1.bias": ("v", None), "layer1.net.4.weight": ("o", "i"), "layer1.net.4.bias": ("v", None),
            "layer1.net.5.weight": ("g", None), "layer1.net.5.bias": ("v", None), "layer2.net.1.net.0.weight": (None, "i"),
            "layer11.upsample.weight": ("to", None), "layer11.upsample.bias": ("v", None),
            "layer11.conv.net.0.weight": ("o", "cat"), "layer11.conv.net.0.bias": ("v", None),
            "layer11.conv.net.1.weight": ("g", None), "layer11.conv.net.1.bias": ("v", None),
            "layer11.conv.net.4.weight": ("o", "i"), "layer11.conv.net.4.bias": ("v", None),
            "layer11.conv.net.5.weight": ("g", None), "layer11.conv.net.5.bias": ("v", None), "layer12.weight": (None, "i")}
    _PAD_BUFFERS = ("layer1.net.1", "layer1.net.5", "layer11.conv.net.1", "layer11.conv.net.5")

    @staticmethod
    def _pad_tensor(t: Tensor, how) -> Tensor:
        o, i = how
        if o == "v":
            return torch.cat([t, torch.zeros_like(t)])
        if o == "g":
            return torch.cat([t, torch.ones_like(t)])
        if o == "o":
            t = torch.cat([t, torch.zeros_like(t)], 0)
        if o == "to":
            t = torch.cat([t, torch.zeros_like(t)], 1)
        if i == "i":
            t = torch.cat([t, torch.zeros_like(t)], 1)
        if i == "cat":
            z = torch.zeros_like(t[:, :16])
            t = torch.cat([t[:, :16], z, t[:, 16:], z], 1)
        return t.contiguous()

    @staticmethod
    def _unpad_grad(gp: Tensor, how) -> Tensor:
        o, i = how
        if o in ("v", "g", "o"):
            gp = gp[:16]
        if o == "to":
            gp = gp[:, :16]
        if i == "i":
            gp = gp[:, :16]
        if i == "cat":
            gp = torch.cat([gp[:, :16], gp[:, 32:48]], 1)
        return gp

    def _pad_params(self, P: Dict[str, Tensor], S: Dict[str, Tensor]):
        if not self.pad16:
            return P, S
        P2 = dict(P)
        for name, how in self._PAD.items():
            P2[name] = self._pad_tensor(P[name].detach(), how)
        S2 = dict(S)
        for bn in self._PAD_BUFFERS:                      # padded running statistics: mean 0, variance 1 (never observed)
            S2[f"{bn}.running_mean"] = torch.cat([S[f"{bn}.running_mean"], torch.zeros_like(S[f"{bn}.running_mean"])])
            S2[f"{bn}.running_var"] = torch.cat([S[f"{bn}.running_var"], torch.ones_like(S[f"{bn}.running_var"])])
            S2[f"{bn}.num_batches_tracked"] = S[f"{bn}.num_batches_tracked"].clone()
        return P2, S2

    def _unpad_buffers(self, S: Dict[str, Tensor], S2: Dict[str, Tensor]):
        if not self.pad16:
            return
        with torch.no_grad():
            for bn in self._PAD_BUFFERS:
                S[f"{bn}.running_mean"].copy_(S2[f"{bn}.running_mean"][:16])
                S[f"{bn}.running_var"].copy_(S2[f"{bn}.running_var"][:16])
                S[f"{bn}.num_batches_tracked"].copy_(S2[f"{bn}.num_batches_tracked"])

    # ------------------------------------------------------------------------------------------------ operands
    def _operands(self, name: str, w: Tensor, kind: str, cop: Optional[int] = None):
        key = (w.data_ptr(), w._version, ops.PARAM_EPOCH[0], self.dtype)
        hit = self._opcache.get(name)
        if hit is None or hit[0] != key:
            wf, wd = ops.weight_prep(w.detach(), kind, self.dtype, cop)
            self._opcache[name] = hit = (key, wf, wd)
        return hit[1], hit[2]

    # ------------------------------------------------------------------------------------------------ forward pieces
    def _bn_relu(self, P, S, p: str, z: Tensor, training: bool) -> Act:
        n, h, w_, c = z.shape
        zv = z.view(1, n * h, w_, c)
        gamma, beta = P[f"{p}.weight"], P[f"{p}.bias"]
        if training:
            stats = ops.instnorm_stats(zv, gamma, beta, self.eps)                  # (4, 1, C): mean, rstd, scale, shift
            with torch.no_grad():
                m = float(n * h * w_)
                mean, rstd = stats[0, 0], stats[1, 0]
                var_unbiased = (1.0 / (rstd * rstd) - self.eps) * (m / max(m - 1.0, 1.0))
                S[f"{p}.running_mean"].mul_(1 - self.momentum).add_(mean, alpha=self.momentum)
                S[f"{p}.running_var"].mul_(1 - self.momentum).add_(var_unbiased, alpha=self.momentum)
                S[f"{p}.num_batches_tracked"].add_(1)
        else:
            rstd = torch.rsqrt(S[f"{p}.running_var"].float() + self.eps)
            scale = gamma.detach() * rstd
            stats = torch.stack([S[f"{p}.running_mean"].float(), rstd, scale,
                                 beta.detach() - S[f"{p}.running_mean"].float() * scale])[:, None].contiguous()
        act = Act(zv, stats, 0.0)
        ops.instnorm_apply(act)
        return act

    def _conv_bn_relu(self, P, S, ctx: Optional[VitalCtx], prefix: str, i: int, srcs: List[Act], training: bool) -> Act:
        w, b = P[f"{prefix}.{i}.weight"], P[f"{prefix}.{i}.bias"]
        co = w.shape[0]
        if w.shape[1] == 1:                                      # first layer: direct convolution of the f32 image
            img = srcs[0]
            n, _, h, w_ = img.shape
            w9, _ = self._first_operand(f"{prefix}.{i}.weight", w)
            z = torch.empty((n, h, w_, co), dtype=self.dtype, device=img.device)
            ops.conv_c1_fwd(img, w9, b, z)
        else:
            wf, _ = self._operands(f"{prefix}.{i}.weight", w, "conv")
            n, h, w_, _ = srcs[0].z.shape
            z = torch.empty((n, h, w_, co), dtype=self.dtype, device=w.device)
            ops.conv_gemm(srcs, wf, b, grid=(h, w_), in_stride=1, taps=TAPS3, dsts=[z], dst_cols=[co])
        act = self._bn_relu(P, S, f"{prefix}.{i + 1}", z, training)
        out = Act(act.a.view(n, h, w_, co), None, 1.0)            # what the consumers read
        if ctx is not None:
            ctx.layers.append(_Layer(prefix, i, [] if w.shape[1] == 1 else srcs, act, (n, h, w_, co)))
        return out

    def _first_operand(self, name: str, w: Tensor):
        key = (w.data_ptr(), w._version, ops.PARAM_EPOCH[0], "c1")
        hit = self._opcache.get(name)
        if hit is None or hit[0] != key:
            w9, _ = ops.weight_prep(w.detach(), "conv", torch.float32, want_dgrad=False)      # [9][CO][1] f32
            self._opcache[name] = hit = (key, w9, None)
        return hit[1], None

    def _double_conv(self, P, S, ctx, prefix: str, srcs, training: bool) -> Act:
        a = self._conv_bn_relu(P, S, ctx, prefix, 0, srcs, training)
        return self._conv_bn_relu(P, S, ctx, prefix, 4, [a], training)

    def _conv_transpose(self, P, ctx, prefix: str, src: Act) -> Act:
        w, b = P[f"{prefix}.weight"], P[f"{prefix}.bias"]        # (CI, CO, 2, 2)
        wf, _ = self._operands(f"{prefix}.weight", w, "convT")
        n, h, w_, _ = src.z.shape
        co = w.shape[1]
        u = torch.empty((n, 2 * h, 2 * w_, co), dtype=self.dtype, device=w.device)
        if co % 32 == 0:
            ops.conv_gemm([src], wf.view(1, 4 * co, wf.shape[2]), b, grid=(h, w_), in_stride=1, taps=[(0, 0, 0)], dsts=[u],
                          dst_cols=[co], out_stride=2, n_cols=4 * co, parity_cols=co)
        else:
            for dy in range(2):
                for dx in range(2):
                    ops.conv_gemm([src], wf, b, grid=(h, w_), in_stride=1, taps=[(0, 0, dy * 2 + dx)], dsts=[u],
                                  dst_cols=[co], out_stride=2, out_off=(dy, dx))
        out = Act(u, None, 1.0)
        if ctx is not None:
            ctx.ups.append((prefix, src, out))
        return out

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, P: Dict[str, Tensor], S: Dict[str, Tensor], img: Tensor, training: bool, keep: bool = True):
        """P: parameters, S: BatchNorm buffers (updated in place when training), img (N, 1, H, W) f32 -> logits, ctx."""
        assert img.dtype == torch.float32 and img.is_cuda and img.shape[1] == 1
        n, _, h, w_ = img.shape
        if h % 32 or w_ % 32:
            raise NotImplementedError("image sides must be multiples of 32 (five poolings; the reference pads otherwise)")
        img = img.contiguous()
        S_user = S
        P, S = self._pad_params(P, S)
        ctx = VitalCtx(img=img) if keep else None
        skips = [self._double_conv(P, S, ctx, "layer1.net", [img], training)]
        for k in range(2, 7):
            y, idx = ops.maxpool2_fwd(skips[-1].z)
            if ctx is not None:
                ctx.pools.append(idx)
            skips.append(self._double_conv(P, S, ctx, f"layer{k}.net.1.net", [Act(y, None, 1.0)], training))
        out = skips.pop()
        for k in range(7, 12):
            skip = skips.pop()
            up = self._conv_transpose(P, ctx, f"layer{k}.upsample", out)
            out = self._double_conv(P, S, ctx, f"layer{k}.conv.net", [skip, up], training)
        w = P["layer12.weight"]
        wf, _ = self._operands("layer12.weight", w, "conv", cop=32)
        bias = torch.zeros(32, dtype=torch.float32, device=w.device)
        bias[: self.num_classes] = P["layer12.bias"].detach()
        logits = torch.empty((n, self.num_classes, h, w_), dtype=torch.float32, device=img.device)
        ops.conv_gemm([out], wf, bias, grid=(h, w_), in_stride=1, taps=[(0, 0, 0)], dsts=[logits], dst_cols=[32],
                      out_nchw=True, n_cols=32)
        if training:
            self._unpad_buffers(S_user, S)
        if ctx is not None:
            ctx.last = out
            ctx.P = P                         # the (padded) parameters the backward must use
        return logits, ctx

    # ------------------------------------------------------------------------------------------------ backward
    def _layer_bwd(self, P, G, ctx: VitalCtx, layer: _Layer, g: Tensor, want_dsrc: bool) -> List[Optional[Tensor]]:
        """g: dL/d(ReLU output) (N, H, W, C), overwritten with dL/dz.  -> gradient per source (None for the image)."""
        n, h, w_, co = layer.shape
        bn = f"{layer.prefix}.{layer.i + 1}"
        gv = g.view(1, n * h, w_, co)
        ops.instnorm_lrelu_bwd(gv, layer.out, P[f"{bn}.weight"], G[f"{bn}.weight"], G[f"{bn}.bias"], None)
        # the conv bias sits in front of a BatchNorm: its gradient (sum of dz) is identically zero and stays 0
        wname = f"{layer.prefix}.{layer.i}.weight"
        w = P[wname]
        if not layer.srcs:
            dw9 = torch.zeros((9, co), dtype=torch.float32, device=g.device)
            ops.conv_c1_wgrad(ctx.img, g, dw9)
            ops.grad_unprep(dw9.view(9, co, 1), G[wname], "conv", accumulate=True)
            return [None]
        ci = w.shape[1]
        dwk = torch.zeros((9, co, ci), dtype=torch.float32, device=g.device)
        ops.conv_wgrad(layer.srcs, g, dwk, grid=(h, w_), in_stride=1, z_stride=1, taps=TAPS3_W, n_cols=co)
        ops.grad_unprep(dwk, G[wname], "conv", accumulate=True)
        if not want_dsrc:
            return [None] * len(layer.srcs)
        _, wd = self._operands(wname, w, "conv")
        cols = [s.z.shape[3] for s in layer.srcs]
        gz = Act(g, None, 1.0)
        if len(cols) == 1 or cols[0] % 32 == 0:
            dsts = [torch.empty((n, h, w_, c), dtype=self.dtype, device=g.device) for c in cols]
            ops.conv_gemm([gz], wd, None, grid=(h, w_), in_stride=1, taps=TAPS3_D, dsts=dsts, dst_cols=cols)
            return dsts
        both = torch.empty((n, h, w_, ci), dtype=self.dtype, device=g.device)     # a 16-channel split point: split by copy
        ops.conv_gemm([gz], wd, None, grid=(h, w_), in_stride=1, taps=TAPS3_D, dsts=[both], dst_cols=[ci])
        return [both[..., :cols[0]].contiguous(), both[..., cols[0]:].contiguous()]

    def backward(self, P: Dict[str, Tensor], G: Dict[str, Tensor], ctx: VitalCtx, dlogits: Tensor):
        """Accumulates every parameter gradient into G (float32, reference layouts)."""
        if self.pad16:                               # gradients of the padded tensors, sliced back at the end
            G_user, P = G, ctx.P
            G = {k: (torch.zeros_like(P[k], dtype=torch.float32) if k in self._PAD else v) for k, v in G_user.items()}
        dt = self.dtype
        last = ctx.last
        n, h, w_, c_last = last.z.shape
        dl = ops.nchw_f32_to_nhwc(dlogits.contiguous(), dt, cp=32)
        G["layer12.bias"] += dlogits.sum((0, 2, 3))
        dwk = torch.zeros((1, 32, c_last), dtype=torch.float32, device=dl.device)
        ops.conv_wgrad([last], dl, dwk, grid=(h, w_), in_stride=1, z_stride=1, taps=[(0, 0, 0, 0, 0)], n_cols=32)
        ops.grad_unprep(dwk, G["layer12.weight"], "conv", accumulate=True)
        _, wd = self._operands("layer12.weight", P["layer12.weight"], "conv", cop=32)
        g = torch.empty_like(last.z)
        ops.conv_gemm([Act(dl, None, 1.0)], wd, None, grid=(h, w_), in_stride=1, taps=[(0, 0, 0)], dsts=[g], dst_cols=[c_last])
        layers = list(ctx.layers)                   # forward order: 6 double convs down (12 layers), 5 up (10 layers)
        d_skip: List[Optional[Tensor]] = [None] * 6
        # ---- decoder
        for k in range(11, 6, -1):                  # layer11 ... layer7
            conv2, conv1 = layers.pop(), layers.pop()
            (g1,) = self._layer_bwd(P, G, ctx, conv2, g, True)
            d_s, d_up = self._layer_bwd(P, G, ctx, conv1, g1, True)
            level = 11 - k                           # skip of layer11 is x1 (level 0), of layer7 x5 (level 4)
            d_skip[level] = d_s
            prefix, src, _ = ctx.ups.pop()
            w = P[f"{prefix}.weight"]
            ci, co = w.shape[0], w.shape[1]
            sh, sw = src.z.shape[1:3]
            G[f"{prefix}.bias"] += d_up.float().sum((0, 1, 2))
            dwk = torch.zeros((4, co, ci), dtype=torch.float32, device=g.device)
            ops.conv_wgrad([src], d_up, dwk, grid=(sh, sw), in_stride=1, z_stride=2,
                           taps=[(0, 0, dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)], n_cols=co)
            ops.grad_unprep(dwk, G[f"{prefix}.weight"], "convT", accumulate=True)
            _, wdT = self._operands(f"{prefix}.weight", w, "convT")
            g = torch.empty_like(src.z)
            ops.conv_gemm([Act(d_up, None, 1.0)], wdT, None, grid=(sh, sw), in_stride=2, taps=CONVT_TAPS, dsts=[g],
                          dst_cols=[ci])
        # ---- encoder: g = gradient of x6 (bottom); every level above adds its skip gradient to the pooling's
        for level in range(5, 0, -1):
            conv2, conv1 = layers.pop(), layers.pop()
            (g1,) = self._layer_bwd(P, G, ctx, conv2, g, True)
            (gp,) = self._layer_bwd(P, G, ctx, conv1, g1, True)
            g = ops.maxpool2_bwd(gp, ctx.pools.pop())
            g += d_skip[level - 1]
        conv2, conv1 = layers.pop(), layers.pop()
        (g1,) = self._layer_bwd(P, G, ctx, conv2, g, True)
        self._layer_bwd(P, G, ctx, conv1, g1, False)
        if self.pad16:
            for name, how in self._PAD.items():
                G_user[name] += self._unpad_grad(G[name], how)
