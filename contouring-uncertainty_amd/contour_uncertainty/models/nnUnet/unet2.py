"""``contour_uncertainty.models.nnUnet.unet2`` -- drop-in ``UNet`` / ``ConfidenceNet`` running on the HIP kernels.

Mirrors the constructor, attribute names, parameter names/shapes and return values of the reference classes
(reference contour_uncertainty/models/nnUnet/unet2.py:14-34 ``ConfidenceNet``, :37-208 ``UNet``) so that
``config/task/model/unet2.yaml`` instantiates it unchanged and reference checkpoints load with ``strict=True``.
The ``nn.Conv2d`` / ``nn.InstanceNorm2d`` / ``nn.ConvTranspose2d`` sub-modules are parameter holders only: ``forward``
runs the kernel schedule of :mod:`cu_hip.engine` through one autograd node (hand-written backward), not PyTorch ops.
"""
from __future__ import annotations

import contextlib
import os
from typing import Sequence, Tuple

import torch
from torch import Tensor, nn

from cu_hip import lib as _lib
from cu_hip import ops as _ops
from cu_hip.engine import ConfidenceEngine, UNetEngine


def _dtype_of(name) -> torch.dtype:
    if isinstance(name, torch.dtype):
        return name
    return {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32,
            "float32": torch.float32}[str(name)]


# ------------------------------------------------------------------------------------------------ parameter holders
class ConvLayer(nn.Module):
    """conv3x3 -> InstanceNorm2d(affine) -> LeakyReLU (reference layers.py:167-205); holder of the parameters."""

    def __init__(self, in_channels: int, out_channels: int, stride: int, negative_slope: float):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 3, stride, 1)
        self.norm = nn.InstanceNorm2d(out_channels, affine=True)
        self.lrelu = nn.LeakyReLU(negative_slope, inplace=True)
        self.use_drop_block = False


class ConvBlock(nn.Module):
    """reference layers.py:208-238"""

    def __init__(self, in_channels: int, out_channels: int, stride: int, negative_slope: float):
        super().__init__()
        self.conv1 = ConvLayer(in_channels, out_channels, stride, negative_slope)
        self.conv2 = ConvLayer(out_channels, out_channels, 1, negative_slope)


class UpsampleBlock(nn.Module):
    """reference layers.py:389-438"""

    def __init__(self, in_channels: int, out_channels: int, stride: int, negative_slope: float):
        super().__init__()
        self.transp_conv = nn.ConvTranspose2d(in_channels, out_channels, stride, stride, 0, 0, bias=False)
        self.conv_block = ConvBlock(2 * out_channels, out_channels, 1, negative_slope)
        self.attention = False


class OutputBlock(nn.Module):
    """reference layers.py:441-463"""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False)


def _flatten_params(module: nn.Module, names: Sequence[str]):
    """Re-home the listed parameters in one flat float32 buffer (fused Adam / bucketed all-reduce work on slices)."""
    params = dict(module.named_parameters())
    plist = [params[n] for n in names]
    total = sum(p.numel() for p in plist)
    dev = plist[0].device
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    off = 0
    for p in plist:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        off += n
    return flat


def _is_flat(plist) -> bool:
    ptr = plist[0].data_ptr()
    for p in plist:
        if p.data_ptr() != ptr or not p.is_contiguous():
            return False
        ptr += p.numel() * 4
    return True


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "UNet", slot, x: Tensor, *params: Tensor):
        P = dict(zip(module._pnames, params))
        # (ctx.needs_input_grad only mirrors requires_grad of the inputs: under no_grad it is still True for parameters;
        # the slot exists exactly when grad mode was on at the call)
        need = slot is not None and any(ctx.needs_input_grad)
        logits, feats, ectx = module.engine.forward(P, x, module.bottleneck_out, module.training or module.mc_dropout,
                                                    keep=need, fused_head=need and slot.fused)
        ctx.module, ctx.ectx, ctx.slot = module, (ectx if need else None), slot
        ctx.save_for_backward(*params)
        if slot is not None:
            slot.feats_event = ectx.feats_event
        if logits is None:
            # fused head (cu_hip.head runs cu_head_fused_fwd on ectx.head): the logits do not exist; their stand-in in
            # autograd is a stride-0 zero tensor of their shape that only cu_hip.head.dsnt_nll knows how to read
            slot.head = ectx.head
            n, _, h, w_ = x.shape
            logits = _ops.zero_placeholder((n, module.num_classes, h, w_), torch.float32, x.device)
        elif slot is not None:
            slot.fused = False
        if module.bottleneck_out:
            return logits, feats
        return logits

    @staticmethod
    def backward(ctx, dlogits, dfeats=None):
        module: UNet = ctx.module
        params = ctx.saved_tensors
        P = dict(zip(module._pnames, params))
        used = module._used_names
        total = sum(P[n].numel() for n in used)
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        G, off = {}, 0
        for n in used:
            k = P[n].numel()
            G[n] = flat[off:off + k].view(P[n].shape)
            off += k
        module.last_flat_grad = flat
        if module.flat_grad_hook is not None:
            module.flat_grad_hook(flat)
        # the skew head (side mode) leaves dL/dfeats in the slot and hands autograd a stride-0 zero stand-in: the real tensor
        # is still being written on the head's stream, so it is only read behind ops.pending_wait(), where the engine needs
        # it; a dense ``dfeats`` is a genuine extra gradient of the bottleneck (another consumer) and is added there
        side_gin = ctx.slot.take_feats() if ctx.slot is not None else None
        if side_gin is not None:
            extra = dfeats if (dfeats is not None and not all(s_ == 0 for s_ in dfeats.stride())) else None

            def dfeats(extra=extra, side_gin=side_gin):       # called by the engine after its pending_wait()
                return side_gin if extra is None else side_gin + extra
        # the DSNT head may have left dL/dlogits in the engine's layout (cu_hip.head.GradSlot); its stand-in in autograd is
        # a stride-0 zero tensor, and anything else that reached ``dlogits`` is a genuine extra gradient to add
        if ctx.ectx is not None and ctx.ectx.head is not None:
            # fused head: dL/d(mu, Sigma) were left in the slot by dsnt_nll's backward (none = the loss does not depend on
            # the landmarks); a dense gradient on the placeholder means something else read the stand-in logits
            if dlogits is not None and not all(s == 0 for s in dlogits.stride()) and bool(dlogits.ne(0).any()):
                raise _lib.ContourHipError("UNet.fused_head(): the logits placeholder was used outside cu_hip.head.dsnt_nll")
            hg = ctx.slot.take_head()
            if hg is None:
                n, k = ctx.ectx.img.shape[0], module.num_classes
                z0 = torch.zeros((n, k, 3), dtype=torch.float32, device=params[0].device)
                aux0 = torch.zeros((n, k, 8), dtype=torch.float32, device=params[0].device)
                aux0[..., 0] = 1e30          # log-sum-exp far above any logit: softmax weights 0 (never 0 * inf)
                aux0[..., 1] = 1.0
                hg = (aux0, z0[..., :2].contiguous(), z0, True)
            with _lib.device_guard(hg[0]):
                module.engine.backward(P, G, ctx.ectx, None, dfeats, head_grads=hg)
            ctx.ectx = None
            return (None, None, None) + tuple(G.get(n) for n in module._pnames)
        dl_nhwc = ctx.slot.take() if ctx.slot is not None else None
        if dl_nhwc is not None and all(s == 0 for s in dlogits.stride()):
            dlogits = None
        with _lib.device_guard(dl_nhwc if dlogits is None else dlogits):
            module.engine.backward(P, G, ctx.ectx, dlogits, dfeats, dl_nhwc=dl_nhwc)
        ctx.ectx = None
        return (None, None, None) + tuple(G.get(n) for n in module._pnames)


class UNet(nn.Module):
    """A generic 2-D U-Net that can be instantiated dynamically (reference unet2.py:37-208), MI355X kernels inside.

    Extra keyword (not in the reference): ``compute_dtype`` = "bf16" (default; production) | "f32" (parity mode).
    """

    def __init__(
            self,
            input_shape: Tuple[int],
            output_shape: Tuple[int],
            patch_size: list,
            kernels: list,
            strides: list,
            normalization_layer: str = "instance",
            negative_slope: float = 1e-2,
            deep_supervision: bool = False,
            attention: bool = False,
            drop_block: bool = False,
            residual: bool = False,
            out_seg_bias: bool = False,
            ssn_rank=0,
            bottleneck_out: bool = False,
            compute_dtype="bf16",
    ) -> None:
        super().__init__()
        if len(patch_size) != 2:
            raise NotImplementedError("Only 2D patches are on the MI355X dsnt path")
        for flag, name in ((deep_supervision, "deep_supervision"), (attention, "attention"), (residual, "residual"),
                           (out_seg_bias, "out_seg_bias"), (ssn_rank != 0, "ssn_rank")):
            if flag:
                raise NotImplementedError(f"{name} is not enabled by any task=dsnt-* config and is out of scope "
                                          "(SURVEY.md section 2 row 3)")
        if normalization_layer != "instance":
            raise NotImplementedError("only InstanceNorm is on the dsnt path (unet2.yaml)")
        for k in kernels:
            if tuple(k) != (3, 3):
                raise NotImplementedError("only 3x3 kernels are on the dsnt path (unet2.yaml)")
        self.patch_size = patch_size
        self.dim = 2
        self.in_channels = int(input_shape[0])
        self.num_classes = int(output_shape[0])
        self.attention, self.residual, self.out_seg_bias = attention, residual, out_seg_bias
        self.negative_slope = negative_slope
        self.deep_supervision = deep_supervision
        self.norm = normalization_layer + "norm2d"
        st = [int(s[0]) for s in strides]
        self.filters = [min(2 ** (5 + i), 480) for i in range(len(st))]
        f = self.filters
        self.input_block = ConvBlock(self.in_channels, f[0], st[0], negative_slope)
        self.downsamples = nn.ModuleList(
            [ConvBlock(f[i], f[i + 1], st[i + 1], negative_slope) for i in range(len(st) - 2)])
        self.bottleneck = ConvBlock(f[-2], f[-1], st[-1], negative_slope)
        self.upsamples = nn.ModuleList(
            [UpsampleBlock(ci, co, s, negative_slope)
             for ci, co, s in zip(f[1:][::-1], f[:-1][::-1], st[1:][::-1])])
        self.output_block = OutputBlock(f[0], self.num_classes)
        self.ssn_rank = ssn_rank
        self.bottleneck_out = bottleneck_out
        self.confidence_net = ConfidenceNet  # reference unet2.py:172: handle used by DSNTSkew
        # never used on the dsnt path but part of the reference state_dict (unet2.py:174, 262-273)
        self.deep_supervision_heads = nn.ModuleList(
            [OutputBlock(f[i + 1], self.num_classes) for i in range(len(self.upsamples) - 1)])
        self.apply(self.initialize_weights)

        self.engine = UNetEngine(self.in_channels, self.num_classes, st, f, negative_slope, 1e-5,
                                 _dtype_of(compute_dtype))
        self.drop_block = drop_block
        self.mc_dropout = False     # True: Dropout2d stays active in eval mode (reference utils/mcdropout.py:89-137)
        if drop_block:
            # reference unet2.py:129-136 (bottleneck) and :302: `len(in_channels) - i <= 2` where in_channels is
            # filters[:-1] (unet2.py:123, nd + 1 entries for nd downsample blocks), i.e. ONLY the last downsample block;
            # both ConvLayers of a block get the Dropout2d (layers.py:231-232)
            nd = len(self.downsamples)
            blocks = [f"downsamples.{i}" for i in range(nd) if (nd + 1) - i <= 2] + ["bottleneck"]
            for b in blocks:
                for li in ("conv1", "conv2"):
                    self.engine.drop_layers.add(f"{b}.{li}")
                    getattr(self.get_submodule(b), li).use_drop_block = True
        self._pnames = [n for n, _ in self.named_parameters()]
        self._used_names = [n for n in self._pnames if not n.startswith("deep_supervision_heads")]
        self._flat = None
        self.last_flat_grad = None
        self.flat_grad_hook = None     # called with the flat gradient buffer at the start of every backward (DDP)
        self._fuse_head = False        # set inside ``with model.fused_head():``

    def initialize_weights(self, module: nn.Module) -> None:
        """Kaiming-normal(a=negative_slope) weights, zero conv biases (reference unet2.py:309-314)."""
        if isinstance(module, (nn.Conv2d, nn.ConvTranspose2d)):
            module.weight = nn.init.kaiming_normal_(module.weight, a=self.negative_slope)
            if module.bias is not None:
                module.bias = nn.init.constant_(module.bias, 0)

    def set_compute_dtype(self, dtype):
        self.engine.dtype = _dtype_of(dtype)
        self.engine._opcache.clear()

    def _apply(self, fn, recurse=True):
        self._plists = None          # .to() / .float() / ... may re-home parameters: walk the module tree again
        return super()._apply(fn, recurse)

    def _params(self):
        """(all parameters in ``_pnames`` order, the used ones) -- cached: walking the module tree costs ~1 ms per call.
        ``_apply`` drops the cache; operations that replace Parameter OBJECTS without it (``load_state_dict(assign=True)``,
        ``conv.weight = nn.Parameter(...)``) are caught by ``_param_probe``: the identity of every holder's ``weight`` / ``bias``
        entry, checked on every call (~30 us) -- ADVICE r3."""
        pl = getattr(self, "_plists", None)
        if pl is not None and self._param_probe() != self._plists_probe:
            pl = None
        if pl is None:
            named = dict(self.named_parameters())
            used = set(self._used_names)
            pl = self._plists = ([named[n] for n in self._pnames], [named[n] for n in self._pnames if n in used])
            self._holders = [m for m in self.modules() if m._parameters]
            self._plists_probe = self._param_probe()
        return pl

    def _param_probe(self):
        hs = getattr(self, "_holders", None)
        if hs is None:
            return None
        return [id(p) for m in hs for p in m._parameters.values()]

    def _ensure_flat(self):
        plist = self._params()[1]
        if self._flat is None or not _is_flat(plist) or self._flat.device != plist[0].device:
            self._flat = _flatten_params(self, self._used_names)

    def flat_params(self):
        """(flat parameter buffer, flat gradient buffer of the last backward) over the used parameters."""
        self._ensure_flat()
        return self._flat, self.last_flat_grad

    @contextlib.contextmanager
    def fused_head(self):
        """Inside this block a grad-enabled ``forward`` may return a PLACEHOLDER for the logits (a stride-0 zero tensor of
        their shape): the last ConvLayer's activation, the 1x1 OutputBlock and the DSNT moments then run as one pass over
        that layer's raw output inside ``cu_hip.head.dsnt_nll`` (head_fused.hip), and the logits never exist.  Only
        ``dsnt_nll`` may consume such a tensor (its shape is real, its values are not) -- which is why this is opt-in and
        used by the dsnt tasks' ``_shared_step`` alone.  Not taken (ordinary logits come back) in f32 parity mode, without
        grad, in deterministic mode, or for shapes the kernels do not serve."""
        prev, self._fuse_head = self._fuse_head, True
        try:
            yield self
        finally:
            self._fuse_head = prev

    def forward(self, input_data: Tensor):  # noqa: D102
        _lib.require_gpu()
        if not input_data.is_cuda:
            raise _lib.ContourHipError("UNet.forward needs a device tensor: the HIP path has no CPU fallback")
        self._ensure_flat()
        params = self._params()[0]
        if params[0].device != input_data.device:
            raise _lib.ContourHipError(f"input on {input_data.device}, parameters on {params[0].device}")
        from cu_hip.head import GradSlot
        slot = GradSlot(self.engine.dtype) if torch.is_grad_enabled() else None
        if slot is not None and self._fuse_head and input_data.dim() == 4:
            slot.fused = self.engine.head_fusable(input_data.shape[0], input_data.shape[2], input_data.shape[3])
        with _lib.device_guard(input_data):
            out = _UNetFn.apply(self, slot, input_data.float(), *params)
        if slot is not None:
            (out[0] if isinstance(out, tuple) else out)._cu_grad_slot = slot
            if isinstance(out, tuple) and slot.feats_event is not None:
                out[1]._cu_ready_event = slot.feats_event       # ConfidenceNet(side=True) waits for this, not for the decoder
                out[1]._cu_grad_slot = slot                     # ... and hands its input gradient back through the slot
        return out


class _ConfidenceFn(torch.autograd.Function):
    """``ready`` (an event recorded when ``feats`` exists) selects the SIDE mode: the head's launches go to a stream of its own
    (the module's), behind that event only -- beside the U-Net's decoder in the forward pass and beside its backward on the
    way back, instead of ~25 latency-bound launches in front of them.  The results are handed over through
    ``cu_hip.ops.pending_add`` / ``pending_wait`` (see there for who waits)."""

    @staticmethod
    def forward(ctx, module: "ConfidenceNet", ready, slot, feats: Tensor, *params: Tensor):
        P = dict(zip(module._pnames, params))
        ctx.side = None
        ctx.slot = slot
        if ready is not None:
            side = module._side_stream(feats.device)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                out, ectx = module.engine.forward(P, feats)
                done = torch.cuda.Event()
                done.record(side)
            _ops.pending_add(done)
            ctx.side = side
        else:
            out, ectx = module.engine.forward(P, feats)
        ctx.module, ctx.ectx = module, ectx
        ctx.need_in = feats.requires_grad
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, gout):
        module: ConfidenceNet = ctx.module
        params = ctx.saved_tensors
        P = dict(zip(module._pnames, params))
        total = sum(p.numel() for p in params)
        # side mode only when the bottleneck gradient is wanted (the U-Net's backward then waits for it where it needs it) AND
        # nothing autograd does with the results on the MAIN stream can read them early: the parameter gradients
        # are only stashed when every ``p.grad`` is None (an existing ``p.grad`` -- gradient accumulation,
        # zero_grad(set_to_none=False), GradSync's deferred mode -- makes AccumulateGrad run ``p.grad += G`` at once), and
        # the input gradient goes through the producer's GradSlot with a stride-0 stand-in inside autograd
        slot = ctx.slot
        side = ctx.side if (ctx.need_in and slot is not None and slot.feats_grad is None
                            and all(p.grad is None for p in module._params())
                            and not torch.cuda.is_current_stream_capturing()) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(gout.device))
            gout.record_stream(side)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            flat = torch.zeros(total, dtype=torch.float32, device=gout.device)
            G, off = {}, 0
            for n, p in zip(module._pnames, params):
                G[n] = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
            module.last_flat_grad = flat
            with _lib.device_guard(gout):
                gin = module.engine.backward(P, G, ctx.ectx, gout, ctx.need_in)
            if side is not None:
                done = torch.cuda.Event()
                done.record(side)
                _ops.pending_add(done)
        ctx.ectx = None
        if side is not None:
            slot.feats_grad = gin
            gin = _ops.zero_placeholder(gin.shape, gin.dtype, gin.device)
        return (None, None, None, gin) + tuple(G[n] for n in module._pnames)


class ConfidenceNet(nn.Module):
    """Bottleneck (N, 480, 2, 2) -> (N, output_size)  (reference unet2.py:14-34; hard-codes 480 channels, 2x2)."""

    def __init__(self, output_size, compute_dtype="bf16"):
        super().__init__()
        self.model = nn.Sequential(
            nn.Conv2d(480, 128, kernel_size=3, stride=1, padding=1),
            nn.ReLU(),
            nn.Conv2d(128, 128, kernel_size=3, stride=1, padding=1),
            nn.ReLU(),
            nn.Conv2d(128, 128, kernel_size=3, stride=1, padding=1),
            nn.ReLU(),
            nn.Flatten(),
            nn.Linear(128 * 2 * 2, output_size),
        )
        self.engine = ConfidenceEngine(_dtype_of(compute_dtype))
        self._pnames = [n for n, _ in self.named_parameters()]
        self._flat = None
        self.last_flat_grad = None
        self._side = None
        self.side_enabled = os.environ.get("CONTOUR_SKEW_SIDE", "1") == "1"

    def _side_stream(self, device):
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device)
        return self._side

    def set_compute_dtype(self, dtype):
        self.engine.dtype = _dtype_of(dtype)
        self.engine._opcache.clear()

    def _apply(self, fn, recurse=True):
        self._plist = None
        return super()._apply(fn, recurse)

    def _params(self):
        pl = getattr(self, "_plist", None)
        if pl is not None and [id(p) for m in self._holders for p in m._parameters.values()] != self._plist_probe:
            pl = None         # a Parameter object was replaced without _apply (load_state_dict(assign=True), ...): ADVICE r3
        if pl is None:
            named = dict(self.named_parameters())
            pl = self._plist = [named[n] for n in self._pnames]
            self._holders = [m for m in self.modules() if m._parameters]
            self._plist_probe = [id(p) for m in self._holders for p in m._parameters.values()]
        return pl

    def _ensure_flat(self) -> bool:
        """True when the parameters had to be re-homed by this call (copies enqueued on the current stream just now)"""
        plist = self._params()
        if self._flat is None or not _is_flat(plist) or self._flat.device != plist[0].device:
            self._flat = _flatten_params(self, self._pnames)
            return True
        return False

    def flat_params(self):
        self._ensure_flat()
        return self._flat, self.last_flat_grad

    def forward(self, x, side: bool = False):
        """``side=True`` (the dsnt-skew training step): run beside the U-Net on a stream of this module's own when ``x`` is the
        bottleneck a grad-enabled ``UNet.forward`` just returned.  The CALLER then owes a ``cu_hip.ops.pending_wait()`` before
        anything but ``cu_hip.head.dsnt_nll`` reads the result (``dsnt_nll`` does it itself)."""
        _lib.require_gpu()
        if x.shape[1] != 480 or x.shape[2] != 2 or x.shape[3] != 2:
            raise ValueError(f"ConfidenceNet expects a (N, 480, 2, 2) bottleneck (reference unet2.py:22,29), got "
                             f"{tuple(x.shape)}")
        moved = self._ensure_flat()
        params = self._params()
        ready = getattr(x, "_cu_ready_event", None)
        slot = getattr(x, "_cu_grad_slot", None)
        # the side stream runs behind `ready` only, i.e. behind what the current stream held when the bottleneck was produced:
        # parameters written later than that (re-homed just now) are not ordered before it -> this call stays on the current stream
        if moved or not (side and self.side_enabled and torch.is_grad_enabled() and x.dtype == torch.float32
                         and not torch.cuda.is_current_stream_capturing()):
            ready = None
        with _lib.device_guard(x):
            return _ConfidenceFn.apply(self, ready, slot if ready is not None else None, x.float(), *params)
