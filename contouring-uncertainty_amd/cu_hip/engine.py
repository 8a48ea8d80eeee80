"""Kernel schedule of the dynamic nnU-Net-style U-Net (forward + hand-written backward) on the HIP C ABI.

This is the MI355X replacement of ``UNet.forward`` (reference contour_uncertainty/models/nnUnet/unet2.py:177-208) and of
PyTorch autograd's backward through it.  Every ``ConvLayer`` (conv -> InstanceNorm -> LeakyReLU, layers.py:167-205)
becomes: one implicit-GEMM launch that writes the *raw* conv output once, one streaming statistics pass, and the
normalise+activate folded into whichever kernels consume the tensor (next conv, skip concat, transposed conv, weight
gradient).  ``torch.cat`` (layers.py:436) is a two-pointer operand load.  No tensor other than raw conv outputs and
their gradients touches HBM.

Data layout: activations NHWC (N, H, W, C) in ``dtype`` (bfloat16 production / float32 parity); logits NCHW float32.
Parameters stay float32 in the reference's layouts and names; tap-major operand copies are refreshed when a parameter's
version counter changes.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .ops import Act

Tensor = torch.Tensor

TAPS3 = [(kh - 1, kw - 1, kh * 3 + kw) for kh in range(3) for kw in range(3)]
TAPS3_W = [(kh - 1, kw - 1, 0, 0, kh * 3 + kw) for kh in range(3) for kw in range(3)]
# input gradient of a stride-1 3x3 conv: da[y] = sum dz[y + dy] w[kh = 1 - dy]
TAPS3_D = [(dy, dx, (1 - dy) * 3 + (1 - dx)) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]


def _parity_taps(par: int):
    """stride-2 input gradient, one axis: (offset into dz, kernel index) pairs for input parity ``par``."""
    return [(0, 1)] if par == 0 else [(1, 0), (0, 2)]


def _s2_parity_taps():
    """weight tap (kh*3+kw) of (gather tap (u, v) of dz, input parity (a, b)) at [(u*2+v)*4 + a*2+b], -1 = no such tap."""
    k1 = {(par, off): k for par in range(2) for off, k in _parity_taps(par)}
    out = []
    for u in range(2):
        for v in range(2):
            for a in range(2):
                for b in range(2):
                    kh, kw = k1.get((a, u), -1), k1.get((b, v), -1)
                    out.append(kh * 3 + kw if kh >= 0 and kw >= 0 else -1)
    return out


S2_PARITY_TAPS = _s2_parity_taps()
ONE_PASS_S2_DGRAD = True      # bf16: one pass on the lean gather-GEMM (pconv.hip); f32 parity mode keeps four launches


@dataclass
class _ConvRec:
    prefix: str
    srcs: List[Act]
    out: Act
    stride: int
    first: bool = False          # Cin == 1 direct conv
    no_z: bool = False           # first layer without a stored z (out.z aliases the activation): backward = ops.conv_c1_bwd
    drop_mask: Optional[Tensor] = None   # Dropout2d multipliers (N, C) applied between conv and norm


@dataclass
class _UpRec:
    prefix: str
    src: Act
    u: Act


@dataclass
class UNetCtx:
    img: Tensor
    training: bool = False
    keep: bool = True            # False = nothing will be back-propagated: layers keep only their activated output
    convs: Dict[str, _ConvRec] = field(default_factory=dict)
    ups: List[_UpRec] = field(default_factory=list)
    enc: List[Act] = field(default_factory=list)
    bott: Optional[Act] = None
    last: Optional[Act] = None
    n_up: int = 0
    feats_event: Optional[object] = None # recorded when the bottleneck features exist
    raw_prefix: Optional[str] = None     # fused head: this layer keeps only its raw output + statistics
    head: Optional[dict] = None          # fused head: {"act", "w_cls", "w_ch"} of the launch pair in head_fused.hip


class _OnStream:
    """``with`` block that makes ``stream`` current and puts ``prev`` back: torch.cuda.stream() without its device and
    current-stream queries (6-8 us of the host's time per use, ~75 uses per backward pass: tools/bwd_host_profile.py).  The caller
    knows the stream it is on."""
    __slots__ = ("stream", "prev")

    def __init__(self, stream, prev):
        self.stream, self.prev = stream, prev

    def __enter__(self):
        torch.cuda.set_stream(self.stream)
        return self.stream

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


class _WsArena:
    """Bump allocator over one persistent float32 tensor, zeroed once per pass.  The first pass (size still unknown)
    hands out individually zeroed tensors and records the total; from the second pass on every take is a slice."""

    def __init__(self):
        self.buf: Optional[Tensor] = None
        self.off = 0
        self.need = 0

    def begin(self, device):
        if self.buf is None or self.buf.device != device or self.buf.numel() < self.need:
            self.buf = torch.zeros(max(self.need, 1), dtype=torch.float32, device=device) if self.need else None
        elif self.off:
            self.buf[:self.off].zero_()
        self.off = 0
        self.need = 0

    def take(self, n: int, device) -> Tensor:
        n = (n + 63) // 64 * 64
        self.need += n
        if self.buf is not None and self.off + n <= self.buf.numel():
            out = self.buf[self.off:self.off + n]
            self.off += n
            return out
        return torch.zeros(n, dtype=torch.float32, device=device)


class UNetEngine:
    def __init__(self, in_channels: int, num_classes: int, strides: Sequence[int], filters: Sequence[int],
                 negative_slope: float = 1e-2, eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16):
        assert in_channels == 1, "the DSNT path feeds single-channel echo images (SURVEY.md 8b)"
        assert num_classes <= 32
        self.in_channels, self.num_classes = in_channels, num_classes
        self.strides, self.filters = list(strides), list(filters)
        self.slope, self.eps, self.dtype = negative_slope, eps, dtype
        self._opcache: Dict[str, Tuple[int, int, Tensor, Tensor]] = {}
        # called with a parameter-name prefix each time that layer's gradients are final (DDP bucket trigger)
        self.grad_ready_hook: Optional[Callable[[str], None]] = None
        # True: one streaming pass materialises LeakyReLU(InstanceNorm(z)) per layer and every consumer stages a plain
        # operand; False: consumers recompute it in their operand load (less HBM traffic, but VALU-bound thin layers)
        self.materialize = True
        # True: the thin, large layers keep only their raw output + statistics; consumers normalise on load (see _lazy).
        # OFF by default: measured on MI355X (profiles/r03_lazy_act.txt) the in-LDS rewrite costs the streaming kernel and
        # the weight-gradient kernel as much as the apply pass it removes (both are latency-bound per tile, not HBM-bound)
        self.lazy_act = os.environ.get("CONTOUR_LAZY_ACT", "0") == "1"
        # True: the first layer (Cin = 1) derives its InstanceNorm statistics from moments of the image and writes z and a in one pass
        # (cu_conv_c1_fwd_norm) instead of conv -> statistics pass over z -> apply pass
        self.first_fused = os.environ.get("CONTOUR_FIRST_FUSED", "1") == "1"
        # True (with first_fused): the first layer never stores z; its whole backward (norm backward + the 9 x CO weight
        # gradient) is two passes over dL/da that recompute z from the image (cu_conv_c1_bwd) -- no z, no dz
        self.first_no_z = os.environ.get("CONTOUR_FIRST_NO_Z", "1") == "1"
        # True: on maps of <= 64 pixels (8x8 and below) the split-K finish pass of the convolution / input-gradient launch
        # carries the layer's InstanceNorm + LeakyReLU forward / backward (no separate, latency-bound norm launch)
        self.small_norm = os.environ.get("CONTOUR_SMALL_NORM", "1") == "1"
        self.small_norm_px = int(os.environ.get("CONTOUR_SMALL_NORM_PX", "64"))     # largest map (pixels per image) that takes it
        # True: when the caller asks for it (UNet.fused_head(), i.e. the dsnt tasks' training step), the last ConvLayer's
        # InstanceNorm + LeakyReLU, the 1x1 OutputBlock and the DSNT moments run as ONE pass over that layer's raw output, and
        # the backward of all three (+ the reduction pass of that layer's norm backward) as one more (head_fused.hip)
        self.fused_head = os.environ.get("CONTOUR_FUSED_HEAD", "1") == "1"
        self._head_parts: Optional[Tensor] = None
        self._bwd_done = set()
        # True: InstanceNorm + LeakyReLU forward (statistics + materialise) and backward (reduce + apply) each run as ONE
        # resident-chunk launch that reads every tensor once (norm.hip); False: the two-pass kernels
        self.fused_norm = True
        self.fused_stats = os.environ.get("CONTOUR_FUSED_STATS", "1") != "0"
        self.fused_norm_bwd = os.environ.get("CONTOUR_FUSED_NORM_BWD", "1") != "0"
        # True (CONTOUR_DETERMINISTIC=1): every f32 sum of the step has a fixed order and a single adder -- weight gradients
        # with one pixel split, InstanceNorm sums by one workgroup per image, dgamma / dbeta by a finish pass, no epilogue
        # fusions: bit-identical gradients run to run, at a fraction of the speed
        self.deterministic = os.environ.get("CONTOUR_DETERMINISTIC", "0") == "1"
        self._det_ws: Optional[Tensor] = None
        # True: on maps of <= 1024 pixels the norm backward leaves dgamma / dbeta as per-image planes (no same-address atomics:
        # 17 of 31 us of the 16^2 x 480 launch) and ONE launch at the end of the backward adds the images for every layer
        self.param_parts = os.environ.get("CONTOUR_PARAM_PARTS", "1") == "1"
        self._pg_items: list = []
        self._pg_table = None
        self._given_sums: Dict[str, Tensor] = {}       # layer prefix -> norm-backward sums gathered by the producer of its g
        self._producer: Dict[int, str] = {}            # id(Act) of a layer's output -> its prefix (valid for one step)
        # Dropout2d(p=0.5) between conv and norm in the listed ConvLayers (reference unet2.py:129-136,302: the last
        # downsample block and the bottleneck when task.model.drop_block=True); active in training mode only
        self.drop_layers: set = set()
        self.drop_p = 0.5
        self.drop_mask_fn: Optional[Callable[[str, int, int, torch.device], Tensor]] = None   # tests inject masks
        self.debug: Optional[Dict[str, Tensor]] = None    # tests/tools: set to {} to capture per-layer gradients (NHWC)
        # tests: keep the last forward's context in ``_last_ctx`` (every layer's raw output, statistics and activation) WITHOUT
        # changing which kernels run -- ``debug`` switches the fused first layer / head / small-map paths off, this does not
        self.keep_ctx = False
        self._prep_state = None     # (pointer key, item tables, early names, names, copies) of the batched operand preparation
        self._prep_pending = None   # (event, stream that must wait for it, names it does not cover): see _prep_all
        self.prep_overlap = os.environ.get("CONTOUR_PREP_OVERLAP", "1") != "0"
        self.c1_bwd_main = os.environ.get("CONTOUR_C1_BWD_MAIN", "1") != "0"
        self.wgrad_after_dgrad = os.environ.get("CONTOUR_WGRAD_AFTER_DGRAD", "0") == "1"
        self.red_stream = os.environ.get("CONTOUR_RED_STREAM", "1") != "0"       # slab sums on a third stream (else: weight-gradient stream)
        self._main_stream = None     # the stream a backward pass runs on, looked up once per pass (None outside a pass)
        self._on_side = False        # inside a weight-gradient task on the side stream
        self.join_probe: Optional[list] = None      # tools: a list collects (main, side, reduction) events at the end of each backward
        self._dwk_ws = None                         # partial-tile scratch of the weight gradients (two buffers)
        self._red: Optional[torch.cuda.Stream] = None      # stream of the partial tiles' sums / un-preparations
        self._red_done = [None, None]
        self._wg_count = 0
        # workgroups of a weight-gradient launch on the second stream: 192 of the 256 CUs' worth, so that the input-gradient
        # chain beside it finds free CUs (a weight-gradient workgroup owns its CU's whole LDS).  Measured, alternating runs
        # on one box: 14.10 / 14.17 ms at 192 against 14.40 / 14.34 ms at 256 (0 = the library's choice, 256)
        self._wgrad_wgs = int(os.environ.get("CONTOUR_WGRAD_WGS", "192"))
        self._wgrad_wgs_thin = int(os.environ.get("CONTOUR_WGRAD_WGS_THIN", str(self._wgrad_wgs)))
        self._dw9_ws: Optional[Tensor] = None       # first layer's 9 x CO accumulator (zero between uses)
        # InstanceNorm workspaces (atomics targets) of all layers of one pass: slices of ONE arena per direction that
        # is zeroed by one fill at the start of the pass (instead of one memset launch per layer)
        self._arena = {"fwd": _WsArena(), "bwd": _WsArena()}
        # weight gradients (+ their un-preparation) run on a second HIP stream beside the input-gradient chain: the two
        # only meet in dz, and each family's prologue / atomics tail is filled by the other's workgroups
        self.side_wgrad = os.environ.get("CONTOUR_SIDE_WGRAD", "1") != "0"
        self._side: Optional[torch.cuda.Stream] = None
        self._side_priority = int(os.environ.get("CONTOUR_SIDE_PRIORITY", "0"))
        self._side_keep: List[Tensor] = []
        # weight-gradient launches may trail the input-gradient chain by `wgrad_lag` layers (round 4 experiment): issued at once,
        # the weight gradient of a level runs beside the input gradient of the SAME level -- both HBM-bound at 256^2 / 128^2,
        # both MFMA-bound in the middle -- so they mostly take turns; lagging them puts an HBM-bound launch beside an
        # MFMA-bound one.  The operands are kept alive by the queue; the data is final when the launch is queued.
        self.wgrad_lag = int(os.environ.get("CONTOUR_WGRAD_LAG", "0"))
        self._lagq: list = []

    # ------------------------------------------------------------------------------------------ operand copies
    def _operands(self, name: str, w: Tensor, kind: str, cop: Optional[int] = None):
        pend = self._prep_pending
        if pend is not None and name not in pend[2]:        # first use of a copy that the weight-gradient stream refreshes
            pend[1].wait_event(pend[0])
            self._prep_pending = None
        key = (w.data_ptr(), w._version, ops.PARAM_EPOCH[0])
        hit = self._opcache.get(name)
        if hit is not None and hit[0] == key and hit[1].dtype == self.dtype:
            return hit[1], hit[2]
        wf, wd = ops.weight_prep(w.detach(), kind, self.dtype, cop)
        self._opcache[name] = (key, wf, wd)
        return wf, wd

    _PREP_EARLY = ("input_block.", "downsamples.0.")

    def _prep_all(self, P: Dict[str, Tensor]):
        """Refresh the bf16/f32 operand copies of EVERY conv / transposed-conv weight when the parameters changed since the last
        call (optimizer step, load_state_dict, ...): one launch for the two top encoder levels on the current stream and one for
        all the other layers on the weight-gradient stream, beside the first layers of the forward pass (the big copies --
        480 x 480 x 9 -- belong to layers the pass reaches a millisecond later; ``_operands`` makes the pass wait for them)."""
        names = [k for k, v in P.items() if v.dim() == 4 and v.shape[1] > 1 and
                 (k.endswith(".conv.weight") or k.endswith("transp_conv.weight"))]
        if not names:
            return
        ptr_key = (self.dtype, tuple((k, P[k].data_ptr(), tuple(P[k].shape)) for k in names))
        st = self._prep_state
        if st is None or st[0] != ptr_key:
            bufs, groups = {}, []
            overlap = self.prep_overlap and self.side_wgrad and P[names[0]].is_cuda
            early = [k for k in names if k.startswith(self._PREP_EARLY)] if overlap else names
            for grp in (early, [k for k in names if k not in early]):
                entries = []
                for k in grp:
                    w = P[k]
                    kind = "convT" if k.endswith("transp_conv.weight") else "conv"
                    cop = 32 if k == "output_block.conv.weight" else None
                    t, co, ci, _, _ = ops._layout(w.shape, kind)
                    cp = cop or co
                    wf = torch.empty((t, cp, ci), dtype=self.dtype, device=w.device)
                    wd = torch.empty((t, ci, cp), dtype=self.dtype, device=w.device)
                    bufs[k] = (wf, wd)
                    entries.append((w.detach(), wf, wd, kind, cop))
                if entries:
                    table, blocks = ops.prep_table(entries, P[names[0]].device)
                    groups.append((table, blocks, len(entries)))
                else:
                    groups.append(None)
            st = self._prep_state = (ptr_key, groups, set(early), names, bufs)
        ver_key = (ops.PARAM_EPOCH[0], tuple(P[k]._version for k in names))
        if getattr(self, "_prep_ver", None) == (ptr_key, ver_key):
            return
        first, late = st[1]
        if first is not None:
            ops.weight_prep_batch(first[0], first[2], first[1], self.dtype)
        if late is not None:
            dev = P[names[0]].device
            main = torch.cuda.current_stream(dev)
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(dev, priority=self._side_priority)
            self._side.wait_stream(main)         # the optimizer's update, and every earlier reader of the copies
            with torch.cuda.stream(self._side):
                ops.weight_prep_batch(late[0], late[2], late[1], self.dtype)
                ev = torch.cuda.Event()
                ev.record(self._side)
            self._prep_pending = (ev, main, st[2])
        self._prep_ver = (ptr_key, ver_key)
        for k in names:
            wf, wd = st[4][k]
            self._opcache[k] = ((P[k].data_ptr(), P[k]._version, ops.PARAM_EPOCH[0]), wf, wd)

    # ------------------------------------------------------------------------------------------ forward pieces
    def _conv_layer_fwd(self, P, ctx: UNetCtx, prefix: str, srcs: List[Act], stride: int) -> Act:
        w = P[f"{prefix}.conv.weight"]
        wf, _ = self._operands(f"{prefix}.conv.weight", w, "conv")
        n, sh, sw, _ = srcs[0].z.shape
        oh, ow = sh // stride, sw // stride
        co = w.shape[0]
        if not (stride == 1 and len(srcs) == 1 and self._raw_ok(srcs[0], w.shape[1], co)):
            srcs = [ops.materialized(s_) for s_ in srcs]       # kernels without the normalise-on-load path
        z = torch.empty((n, oh, ow, co), dtype=self.dtype, device=w.device)
        # thin, large layers: the streaming kernel gathers the InstanceNorm statistics in its epilogue (no statistics pass)
        sums = None
        # (thin layers: the streaming kernel, stride 1 only; from 64 channels on: the LDS-DMA kernels, both strides -- the
        # launch reports whether its kernel gathered them)
        fusable = (self.fused_stats and not self.deterministic and self.fused_norm and self.materialize and self.dtype == torch.bfloat16
                   and n * oh * ow >= (1 << 14) and oh * ow > 64 and not (ctx.training and prefix in self.drop_layers))
        if fusable:
            sums = self._arena["fwd"].take(2 * n * co, z.device)
        small = None
        if (not fusable and self.small_norm and self.fused_norm and self.materialize and oh * ow <= self.small_norm_px and co % 32 == 0
                and not (ctx.training and prefix in self.drop_layers)):
            small = (P[f"{prefix}.norm.weight"], P[f"{prefix}.norm.bias"], self.eps, self.slope,
                     torch.empty((4, n, co), dtype=torch.float32, device=z.device), torch.empty_like(z))
        got = ops.conv_gemm(srcs, wf, P[f"{prefix}.conv.bias"], grid=(oh, ow), in_stride=stride, taps=TAPS3, dsts=[z],
                            dst_cols=[co], stat_sums=sums, norm_fwd=small)
        if got:
            if small is not None:      # the launch's finish pass wrote z, the statistics and the activation
                out = Act(z, small[4], self.slope, small[5], None)
            else:
                out = ops.instnorm_fwd_given(z, P[f"{prefix}.norm.weight"], P[f"{prefix}.norm.bias"], self.slope, sums,
                                             P[f"{prefix}.conv.bias"], self.eps,
                                             materialize=not (self._lazy(ctx, z) or prefix == ctx.raw_prefix))
            if not ctx.keep and out.a is not None:
                return Act(out.a, None, 1.0)
            ctx.convs[prefix] = _ConvRec(prefix, srcs, out, stride, drop_mask=None)
            self._producer[id(out)] = prefix
            return out
        mask = None
        if ctx.training and prefix in self.drop_layers:
            if self.drop_mask_fn is not None:
                mask = self.drop_mask_fn(prefix, n, co, z.device).float().contiguous()
            else:
                keep = torch.rand((n, co), device=z.device) >= self.drop_p
                mask = keep.float() / (1.0 - self.drop_p)
            ops.channel_scale(z, mask)
        out = self._norm_act_fwd(P, prefix, z, ctx, raw=prefix == ctx.raw_prefix)
        if not ctx.keep and out.a is not None:
            return Act(out.a, None, 1.0)      # inference: z, the statistics and the layer record die here
        ctx.convs[prefix] = _ConvRec(prefix, srcs, out, stride, drop_mask=mask)
        self._producer[id(out)] = prefix
        return out

    def _lazy(self, ctx: "UNetCtx", z: Tensor) -> bool:
        """True: do NOT materialise LeakyReLU(InstanceNorm(z)) of this layer at production.  The thin,
        large layers (256^2 x 32, 128^2 x 64 channels) are HBM-bound, and the consumers that dominate there -- the streaming
        3x3 kernel, the weight-gradient kernel, the 1x1 head -- normalise + activate the raw tensor in LDS while staging it;
        any other consumer materialises it on first need (``ops.materialized``), so this is a pure speed decision."""
        n, h, w_, c = z.shape
        return (self.lazy_act and ctx.keep and self.debug is None and self.dtype == torch.bfloat16 and self.fused_norm and
                self.materialize and not self.deterministic and c in (32, 64) and n * h * w_ >= (1 << 20))

    def _raw_ok(self, src: Act, ci: int, co: int) -> bool:
        """the 3x3 stride-1 kernels (forward + weight gradient) take this source raw"""
        n, h, w_, _ = src.z.shape
        return src.a is None and src.stats is not None and (ci, co) in ((32, 32), (64, 64)) and n * h * w_ >= (1 << 20) \
            and h % 8 == 0 and w_ % 32 == 0

    def _norm_act_fwd(self, P, prefix: str, z: Tensor, ctx: Optional["UNetCtx"] = None, raw: bool = False) -> Act:
        gamma, beta = P[f"{prefix}.norm.weight"], P[f"{prefix}.norm.bias"]
        if self.fused_norm and self.materialize:
            ws = self._arena["fwd"].take(ops.resident_ws_floats(z.shape[0], z.shape[3]), z.device)
            return ops.instnorm_fwd_fused(z, gamma, beta, self.slope, self.eps, ws=ws, mode=self._norm_mode(),
                                          materialize=not (raw or (ctx is not None and self._lazy(ctx, z))))
        out = Act(z, ops.instnorm_stats(z, gamma, beta, self.eps), self.slope)
        if self.materialize:
            ops.instnorm_apply(out)
        return out

    def _first_layer_fwd(self, P, ctx: UNetCtx, prefix: str, img: Tensor) -> Act:
        w = P[f"{prefix}.conv.weight"]                       # (CO, 1, 3, 3)
        co = w.shape[0]
        key = (w.data_ptr(), w._version, ops.PARAM_EPOCH[0])
        hit = self._opcache.get(prefix)
        if hit is None or hit[0] != key:
            w9, _ = ops.weight_prep(w.detach(), "conv", torch.float32, want_dgrad=False)   # [9][CO][1] f32
            self._opcache[prefix] = (key, w9, None)
        w9 = self._opcache[prefix][1]
        n, _, h, w_ = img.shape
        if self.first_fused and self.fused_norm and self.materialize and self.debug is None:
            # statistics from moments of the image, then z and a in one pass (no pass over z between conv and apply)
            no_z = self.first_no_z and ctx.keep and not self.deterministic and not self.lazy_act and co % 32 == 0
            out = ops.conv_c1_fwd_norm(img, w9, P[f"{prefix}.conv.bias"], P[f"{prefix}.norm.weight"],
                                       P[f"{prefix}.norm.bias"], self.slope, self.eps, self.dtype, keep_z=not no_z)
            if no_z:
                ctx.convs[prefix] = _ConvRec(prefix, [], out, 1, first=True, no_z=True)
                self._producer[id(out)] = prefix
                return out
        else:
            z = torch.empty((n, h, w_, co), dtype=self.dtype, device=img.device)
            ops.conv_c1_fwd(img, w9, P[f"{prefix}.conv.bias"], z)
            out = self._norm_act_fwd(P, prefix, z, ctx)
        if not ctx.keep and out.a is not None:
            return Act(out.a, None, 1.0)
        ctx.convs[prefix] = _ConvRec(prefix, [], out, 1, first=True)
        self._producer[id(out)] = prefix
        return out

    def _block_fwd(self, P, ctx, prefix: str, srcs: List[Act], stride: int) -> Act:
        a = self._conv_layer_fwd(P, ctx, f"{prefix}.conv1", srcs, stride)
        return self._conv_layer_fwd(P, ctx, f"{prefix}.conv2", [a], 1)

    def _convT_fwd(self, P, ctx: UNetCtx, prefix: str, src: Act) -> Act:
        w = P[f"{prefix}.weight"]                             # (CI, CO, 2, 2)
        src = ops.materialized(src)
        wf, _ = self._operands(f"{prefix}.weight", w, "convT")
        n, h, w_, _ = src.z.shape
        co = w.shape[1]
        u = torch.empty((n, 2 * h, 2 * w_, co), dtype=self.dtype, device=w.device)
        # all four output parities in one pass over the source: W[4][CO][CI] viewed as 4*CO GEMM columns
        ops.conv_gemm([src], wf.view(1, 4 * co, wf.shape[2]), None, grid=(h, w_), in_stride=1, taps=[(0, 0, 0)], dsts=[u],
                      dst_cols=[co], out_stride=2, n_cols=4 * co, parity_cols=co)
        out = Act(u, None, 1.0)
        if ctx.keep:
            ctx.ups.append(_UpRec(prefix, src, out))
        return out

    # ------------------------------------------------------------------------------------------ forward
    def head_fusable(self, n: int, h: int, w_: int) -> bool:
        """True when ``forward(fused_head=True)`` will take the fused head for an (n, 1, h, w_) batch."""
        return (self.fused_head and self.fused_norm and self.materialize and not self.deterministic and self.debug is None
                and not self.lazy_act and ops.head_fused_ok(n, h, w_, self.filters[0], self.num_classes, self.dtype))

    def forward(self, P: Dict[str, Tensor], img: Tensor, want_bottleneck: bool, training: bool = False,
                keep: bool = True, fused_head: bool = False):
        """P: parameter name -> float32 device tensor (reference names).  img: (N, 1, H, W) float32.
        keep=False (no gradient will be asked for): every layer drops its raw output and statistics as soon as its
        activated output exists, so predict-time peak memory is the encoder skips + one block, not a training step's."""
        assert img.dtype == torch.float32 and img.is_cuda and img.shape[1] == 1
        img = img.contiguous()
        ctx = UNetCtx(img=img, training=training, keep=keep or self.debug is not None)
        if self.debug is not None or self.keep_ctx:
            self._last_ctx = ctx
        st = self.strides
        fused_head = fused_head and keep and self.head_fusable(img.shape[0], img.shape[2], img.shape[3])
        if fused_head:      # the last ConvLayer feeds nothing but the head: its activation is never materialised
            ctx.raw_prefix = f"upsamples.{len(st) - 2}.conv_block.conv2"
        assert st[0] == 1
        self._arena["fwd"].begin(img.device)
        self._producer.clear()
        self._prep_all(P)
        a = self._first_layer_fwd(P, ctx, "input_block.conv1", img)
        a = self._conv_layer_fwd(P, ctx, "input_block.conv2", [a], 1)
        ctx.enc = [a]
        nd = len(st) - 2
        for i in range(nd):
            a = self._block_fwd(P, ctx, f"downsamples.{i}", [a], st[i + 1])
            ctx.enc.append(a)
        a = self._block_fwd(P, ctx, "bottleneck", [a], st[-1])
        ctx.bott = a
        feats = ops.act_to_nchw_f32(a) if want_bottleneck else None
        if feats is not None and not torch.cuda.is_current_stream_capturing():
            # the bottleneck exists from here on: a consumer on another stream (the skew head, cu_hip ConfidenceNet side mode)
            # waits for this event instead of for the whole decoder that is enqueued behind it
            ctx.feats_event = torch.cuda.Event()
            ctx.feats_event.record()
        up_strides = st[1:][::-1]
        for i, skip in enumerate(reversed(ctx.enc)):
            assert up_strides[i] == 2, "transposed conv kernel = stride = 2 on the dsnt path"
            u = self._convT_fwd(P, ctx, f"upsamples.{i}.transp_conv", a)
            a = self._block_fwd(P, ctx, f"upsamples.{i}.conv_block", [u, skip], 1)
        ctx.n_up = len(ctx.enc)
        # (the 1x1 head's kernels read the materialised activation: the generic kernel's fused load made the head's forward
        # 419 instead of 143 us and its weight gradient 231 instead of 165 us at batch 64 -- profiles/r03_lazy_act.txt)
        w = P["output_block.conv.weight"]
        if fused_head and a.a is None and a.stats is not None:
            # fused head (head_fused.hip): the caller (cu_hip.head) runs cu_head_fused_fwd on the raw output; no logits
            wf, wd = self._operands("output_block.conv.weight", w, "conv", cop=32)
            ctx.last = a
            ctx.head = {"act": a, "w_cls": wf, "w_ch": wd, "k": self.num_classes}
            return None, feats, ctx
        a = ops.materialized(a)
        ctx.last = a
        # 1x1 output conv -> NCHW f32 logits (K planes; GEMM columns padded to 32)
        wf, _ = self._operands("output_block.conv.weight", w, "conv", cop=32)
        n, h, w_, _ = a.z.shape
        logits = torch.empty((n, self.num_classes, h, w_), dtype=torch.float32, device=img.device)
        ops.conv_gemm([a], wf, None, grid=(h, w_), in_stride=1, taps=[(0, 0, 0)], dsts=[logits], dst_cols=[32],
                      out_nchw=True, n_cols=32)
        return logits, feats, ctx

    # ------------------------------------------------------------------------------------------ backward pieces
    def _ready(self, prefix: str):
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(prefix)

    def _dwk(self, shape, device) -> Tensor:
        """The first layer's 9 x CO weight-gradient accumulator (f32 atomics target): a persistent buffer that is zeroed
        once and handed back clean by its un-preparation (read-and-clear).  ``backward`` drops it when a step aborts, so a
        half-accumulated buffer can never leak into the next step."""
        n = 1
        for d in shape:
            n *= d
        ws = self._dw9_ws
        if ws is None or ws.numel() < n or ws.device != device:
            ws = self._dw9_ws = torch.zeros(max(n, 9 * 64), dtype=torch.float32, device=device)
        return ws[:n].view(shape)

    def _norm_mode(self) -> int:
        return ops.NORM_WS_CLEAN | (ops.NORM_DETERMINISTIC if self.deterministic else 0)

    def _wgrad_stream(self, *reads: Tensor):
        """``with`` block whose launches go to the weight-gradient stream, ordered after everything enqueued so far.  The
        tensors it reads are kept alive until ``_join_wgrad`` (no caching-allocator reuse under the side stream)."""
        if not self.side_wgrad or not reads[0].is_cuda:
            return contextlib.nullcontext()
        dev = reads[0].device
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(dev, priority=self._side_priority)     # (priorities: profiles/r04_stream_priority.txt)
        main = self._main_stream if self._main_stream is not None else torch.cuda.current_stream(dev)
        self._side.wait_stream(main)
        self._side_keep.extend(reads)
        return _OnStream(self._side, main)

    def _wgrad_task(self, reads, fn):
        """run ``fn`` (a weight-gradient launch + its slab sums) on the weight-gradient stream: now, or ``wgrad_lag`` tasks later"""
        if (not (self.wgrad_lag or self.wgrad_after_dgrad) or not self.side_wgrad or not reads[0].is_cuda
                or torch.cuda.is_current_stream_capturing()):
            ctx = self._wgrad_stream(*reads)
            with ctx:
                self._on_side = isinstance(ctx, _OnStream)
                try:
                    fn()
                finally:
                    self._on_side = False
            return
        ev = torch.cuda.Event()
        ev.record()
        self._lagq.append((ev, reads, fn))
        if self.wgrad_after_dgrad:          # (issued by _flush_after_dgrad, right behind the layer's input-gradient launch)
            return
        while len(self._lagq) > self.wgrad_lag:
            self._run_lagged()

    def _flush_after_dgrad(self):
        """experiment (CONTOUR_WGRAD_AFTER_DGRAD=1): the layer's weight-gradient launches reach their stream AFTER its input-gradient
        launch reached the main one -- same dependencies (the event was recorded before), other issue order"""
        if self.wgrad_after_dgrad:
            while self._lagq:
                self._run_lagged()

    def _run_lagged(self):
        ev, reads, fn = self._lagq.pop(0)
        dev = reads[0].device
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(dev, priority=self._side_priority)
        self._side.wait_event(ev)
        self._side_keep.extend(reads)
        with torch.cuda.stream(self._side):
            fn()

    def _join_wgrad(self, device):
        while self._lagq:
            self._run_lagged()
        ops.pending_wait()                  # (a side task nobody consumed inside this backward: its results are gradients)
        if self.join_probe is not None:     # tools/join_slack.py: when does each stream finish its share of the pass?
            evs = []
            for st in (torch.cuda.current_stream(device), self._side, self._red):
                ev = torch.cuda.Event(enable_timing=True) if st is not None else None
                if ev is not None:
                    ev.record(st)
                evs.append(ev)
            self.join_probe.append(evs)
        if self._side is not None and self._side_keep:
            main = self._main_stream if self._main_stream is not None else torch.cuda.current_stream(device)
            main.wait_stream(self._side)
            if self._red is not None:
                main.wait_stream(self._red)
        self._side_keep.clear()

    PARTS_FLOATS = 24 << 20       # 96 MiB: 256 slabs of a 64 x 64 x 9 block are 9.4 M floats; the library caps the splits

    def _parts_ws(self, device, k: int = 0) -> Tensor:
        """Scratch of the weight gradients' partial tiles (cu_conv_wgrad_parts): every pixel split of a launch stores its
        tile into a slab of its own and cu_grad_unprep_parts adds the slabs in a fixed order -- no f32 atomics (they ran
        at the chip-wide ~1.3 TB/s atomic rate: 29 us per launch whatever the layer), no zeroing, no read-and-clear
        invariant, and bit-identical weight gradients run to run.  Two buffers per engine, used alternately: layer L's sum
        (on the reduction stream) overlaps layer L+1's weight-gradient launch (on the weight-gradient stream)."""
        ws = self._dwk_ws
        if ws is None or ws[0].device != device or ws[0].numel() < self.PARTS_FLOATS:
            ws = self._dwk_ws = [torch.empty(self.PARTS_FLOATS, dtype=torch.float32, device=device) for _ in range(2)]
            self._red_done = [None, None]
        return ws[k]

    def _wgrad(self, srcs, z: Tensor, shape, grad: Tensor, kind: str, prefix: str, **kw):
        """weight gradient of one layer -> ``grad`` (+=): partial tiles (on the current = weight-gradient stream) + their
        ordered sum and un-preparation (light, LDS-free kernels: on a THIRD stream, so that they run beside the next
        layer's weight-gradient launch, which owns every CU's LDS, instead of in front of it)."""
        k = self._wg_count & 1
        self._wg_count += 1
        ws = self._parts_ws(z.device, k)
        cur = self._side if self._on_side else torch.cuda.current_stream(z.device)
        # (inside a hipGraph capture the sums stay on the weight-gradient stream: ROCm 7.2's capture_end crashed on the
        # stream forked from a forked stream; CONTOUR_GRAPH_THIRD=1 re-enables it for experiments)
        third = self.side_wgrad and self.red_stream and self._side is not None and cur == self._side and \
            (not torch.cuda.is_current_stream_capturing() or os.environ.get("CONTOUR_GRAPH_THIRD") == "1")
        if third:
            if self._red is None or self._red.device != z.device:
                self._red = torch.cuda.Stream(z.device)
            if self._red_done[k] is not None:
                cur.wait_event(self._red_done[k])              # buffer k's previous sum has been read
        if self._wgrad_wgs and third and "splits" not in kw:
            # experiment knob (CONTOUR_WGRAD_WGS=n): cap the weight-gradient launch at ~n workgroups so that it leaves CUs to
            # the input-gradient chain running beside it (its workgroups own a CU's whole LDS)
            co, ci = shape[1], shape[2]
            wgs = self._wgrad_wgs_thin if max(co, ci) <= 64 else self._wgrad_wgs      # (thin = the HBM-bound 256^2 / 128^2 levels)
            kw["splits"] = max(1, wgs // (-(-co // 64) * -(-ci // 64)))
        slabs = ops.conv_wgrad(srcs, z, ws, parts=True, **kw)
        if not third:
            ops.grad_unprep_parts(ws, slabs, shape[1], grad, kind, accumulate=True)
            self._ready(prefix)
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        with _OnStream(self._red, cur):
            self._red.wait_event(ev)
            ops.grad_unprep_parts(ws, slabs, shape[1], grad, kind, accumulate=True)
            self._ready(prefix)
            done = torch.cuda.Event()
            done.record(self._red)
        self._red_done[k] = done

    def _unprep(self, dwk: Tensor, grad: Tensor, kind: str, prefix: str):
        """kernel-layout dWk -> logical gradient, layer by layer (measured in round 1: one batched launch at the end of the
        backward made the un-preparation itself 0.5 ms cheaper and the step 0.8 ms slower: cold accumulators)."""
        ops.grad_unprep(dwk, grad, kind, accumulate=True, clear=True)
        if self.grad_ready_hook is not None and self._red is not None and dwk.is_cuda:
            # the bucket this report may complete also holds the conv-weight gradients of the layers before it, whose slab
            # sums run on the reduction stream: the stream the collective is ordered behind must cover both
            torch.cuda.current_stream(dwk.device).wait_stream(self._red)
        self._ready(prefix)

    def _conv_layer_bwd(self, *a, **k):
        self._conv_layer_bwd_impl(*a, **k)
        self._flush_after_dgrad()

    def _conv_layer_bwd_impl(self, P, G, ctx: UNetCtx, prefix: str, g: Tensor,
                        dsrc: Optional[List[Tuple[Tensor, int]]]):
        """g: dL/d(activated output), overwritten with dL/dz.  dsrc: [(tensor, accumulate)] per source or None."""
        rec = ctx.convs[prefix]
        if self.debug is not None:
            self.debug[f"{prefix}:da"] = g.float().clone()
        # d(conv bias) = sum_p dz is identically zero behind an InstanceNorm (dz has zero mean per (n, c)); the reference
        # accumulates rounding noise there.  G[conv.bias] stays exactly 0 (no kernel work, no atomics contention).
        if rec.no_z:
            # the first layer's whole backward from dL/da: z recomputed from the image, dz never stored (cu_conv_c1_bwd)
            n, oh, ow, co = g.shape
            # on the MAIN stream (round 4): this is the last work of the pass and the input-gradient chain has just ended, while the
            # weight-gradient stream still holds the 256^2 weight gradient of the layer above -- behind it the two launches
            # (94 + 134 us) were an exposed tail; beside it they are free (profiles/r04_c1_bwd_main.txt)
            with (contextlib.nullcontext() if self.c1_bwd_main else self._wgrad_stream(g, ctx.img)):
                sums = self._arena["bwd"].take(2 * n * co, g.device).view(n, co, 2)     # (zeroed with the pass's arena)
                dw9 = self._dwk((9, co), g.device)
                ops.conv_c1_bwd(ctx.img, self._opcache[prefix][1], P[f"{prefix}.conv.bias"], rec.out.stats,
                                P[f"{prefix}.norm.weight"], self.slope, g, sums, dw9, G[f"{prefix}.norm.weight"],
                                G[f"{prefix}.norm.bias"])
                self._unprep(dw9.view(9, co, 1), G[f"{prefix}.conv.weight"], "conv", prefix)
            return
        given = self._given_sums.pop(prefix, None)
        if prefix in self._bwd_done:   # the launch that produced g did this layer's whole norm backward: g IS dL/dz
            self._bwd_done.discard(prefix)
        elif given is not None:        # the launch that produced g already did the reduction pass (tconv epilogue)
            ops.instnorm_bwd_given(g, rec.out, P[f"{prefix}.norm.weight"], G[f"{prefix}.norm.weight"],
                                   G[f"{prefix}.norm.bias"], given)
        elif self.fused_norm:
            ws = self._arena["bwd"].take(ops.resident_ws_floats(g.shape[0], g.shape[3]), g.device)
            parts = self._pgrad_parts(G, prefix, g.shape[0], g.shape[1] * g.shape[2], g.shape[3], g.device)
            if parts is not None:
                ops.instnorm_bwd_fused(g, rec.out, P[f"{prefix}.norm.weight"], parts[0], parts[1], ws,
                                       mode=self._norm_mode() | ops.NORM_PARAM_PARTS)
            else:
                ops.instnorm_bwd_fused(g, rec.out, P[f"{prefix}.norm.weight"], G[f"{prefix}.norm.weight"],
                                       G[f"{prefix}.norm.bias"], ws, mode=self._norm_mode())
        else:
            ops.instnorm_lrelu_bwd(g, rec.out, P[f"{prefix}.norm.weight"], G[f"{prefix}.norm.weight"],
                                   G[f"{prefix}.norm.bias"], None)
        if rec.drop_mask is not None:          # gradient through the Dropout2d that sits between conv and norm
            ops.channel_scale(g, rec.drop_mask)
        if self.debug is not None:
            self.debug[f"{prefix}:dz"] = g.float().clone()
        w = P[f"{prefix}.conv.weight"]
        n, oh, ow, co = g.shape
        if rec.first:
            with self._wgrad_stream(g, ctx.img):
                dw9 = self._dwk((9, co), g.device)      # 9 x CO floats: the first layer keeps its small atomics target
                det_ws = None
                if self.deterministic:
                    if self._det_ws is None or self._det_ws.device != g.device:
                        self._det_ws = torch.empty(1 << 20, dtype=torch.float32, device=g.device)
                    det_ws = self._det_ws
                ops.conv_c1_wgrad(ctx.img, g, dw9, det_ws)
                self._unprep(dw9.view(9, co, 1), G[f"{prefix}.conv.weight"], "conv", prefix)
            return
        ci = w.shape[1]
        if not (rec.stride == 1 and len(rec.srcs) == 1 and self._raw_ok(rec.srcs[0], ci, co)):
            rec.srcs = [ops.materialized(s_) for s_ in rec.srcs]
        srcs_w, stride_w = rec.srcs, rec.stride
        self._wgrad_task([g] + [s_.z for s_ in srcs_w] + [s_.a for s_ in srcs_w if s_.a is not None],
                         lambda: self._wgrad(srcs_w, g, (9, co, ci), G[f"{prefix}.conv.weight"], "conv", prefix, grid=(oh, ow),
                                             in_stride=stride_w, z_stride=1, taps=TAPS3_W, n_cols=co))
        if dsrc is None:
            return
        _, wd = self._operands(f"{prefix}.conv.weight", w, "conv")
        dsts = [d[0] for d in dsrc]
        acc = [d[1] for d in dsrc] + [0]
        cols = [s.z.shape[3] for s in rec.srcs]
        gz = Act(g, None, 1.0)
        sh, sw = rec.srcs[0].z.shape[1:3]
        if rec.stride == 1:
            # single destination = the output gradient of the layer that feeds this one: its norm backward's reduction
            # pass can ride in this launch's epilogue (thin, large layers; no Dropout2d between its conv and norm)
            nb = None
            src = rec.srcs[0] if len(rec.srcs) == 1 else None
            tgt = self._producer.get(id(src)) if src is not None else None
            if (tgt is not None and self.fused_norm_bwd and not self.deterministic and self.fused_norm and self.dtype == torch.bfloat16 and not acc[0]
                    and n * sh * sw >= (1 << 20) and ctx.convs[tgt].drop_mask is None and ctx.convs[tgt].out.stats is not None
                    and not ctx.convs[tgt].no_z):
                nb = (ctx.convs[tgt].out, self._arena["bwd"].take(2 * n * cols[0], g.device))
            full = self._small_norm_bwd(P, G, ctx, src, n, sh * sw, cols[0]) if nb is None and len(dsts) == 1 else None
            got = ops.conv_gemm([gz], wd, None, grid=(sh, sw), in_stride=1, taps=TAPS3_D, dsts=dsts, dst_cols=cols,
                                accum=acc, norm_bwd=nb, norm_bwd_full=full[1] if full else None)
            if got and full:
                self._norm_bwd_taken(full)
            elif got:
                self._given_sums[tgt] = nb[1]
        elif ONE_PASS_S2_DGRAD and self.dtype == torch.bfloat16 and len(dsts) == 1 and cols[0] % 32 == 0:
            # all four input parities in one pass: gather taps = the 2x2 neighbourhood of dz, the weight tap of
            # (gather tap, parity) from S2_PARITY_TAPS (9 of the 16 pairs exist, the others are skipped); dz is read once.
            # (On the tile-generic kernel this form was slower than four parity launches -- 16/9 of the MFMA work; the
            # lean gather-GEMM skips the absent pairs.  tools/conv_bench.py s2dgrad compares the two.)
            full = self._small_norm_bwd(P, G, ctx, rec.srcs[0], n, sh * sw, cols[0])
            got = ops.conv_gemm([gz], wd, None, grid=(sh // 2, sw // 2), in_stride=1,
                                taps=[(u, v, 0) for u in range(2) for v in range(2)], dsts=dsts, dst_cols=cols, out_stride=2,
                                accum=acc, n_cols=4 * cols[0], parity_cols=cols[0], parity_taps=S2_PARITY_TAPS,
                                norm_bwd_full=full[1] if full else None)
            if got and full:
                self._norm_bwd_taken(full)
        else:
            for py in range(2):
                for px in range(2):
                    taps = [(dy, dx, kh * 3 + kw) for dy, kh in _parity_taps(py) for dx, kw in _parity_taps(px)]
                    ops.conv_gemm([gz], wd, None, grid=(sh // 2, sw // 2), in_stride=1, taps=taps, dsts=dsts,
                                  dst_cols=cols, out_stride=2, out_off=(py, px), accum=acc)

    def _pgrad_parts(self, G, prefix: str, n: int, hw: int, c: int, device, register: bool = True):
        """(dgamma planes, dbeta planes), each (N, C) and zero, for a layer whose norm backward may leave per-image parameter
        gradients (``ops.norm_param_parts_ok``), registered for the batched sum at the end of the pass; else None.
        ``register=False`` returns (planes, planes, item) and leaves the registration to the caller: a layer must be in the
        batch ONCE (its work items read-modify-write the same gradient without atomics)."""
        # (with a gradient exchange attached, a layer's "gradients final" report covers its norm parameters too: they must be
        #  final when the layer reports, not at the end of the pass)
        if not self.param_parts or self.deterministic or self.grad_ready_hook is not None or not ops.norm_param_parts_ok(n, hw):
            return None
        buf = self._arena["bwd"].take(2 * n * c, device)
        gp, bp = buf[:n * c], buf[n * c:2 * n * c]
        item = (gp, bp, G[f"{prefix}.norm.weight"], G[f"{prefix}.norm.bias"], n, c)
        if not register:
            return gp, bp, item
        self._pg_items.append(item)
        return gp, bp

    def _pgrad_finish(self, device):
        """one launch: dgamma / dbeta of every registered layer += sum of its per-image planes"""
        items, self._pg_items = self._pg_items, []
        if not items:
            return
        # the destinations are views of the step's flat gradient buffer, which is NEW every step: the table stores their offsets
        # from its base, so it is built once (a table rebuilt per step is a pageable host-to-device copy that makes the host wait
        # for everything enqueued before it: measured +1.4 ms per step)
        assert len({it[2].data_ptr() for it in items}) == len(items), "a layer twice in one batch: its two work items would race"
        st0 = items[0][2].untyped_storage()
        base = st0.data_ptr() if all(t.untyped_storage().data_ptr() == st0.data_ptr() for it in items for t in it[2:4]) else 0
        key = tuple((a.data_ptr(), b.data_ptr(), g_.data_ptr() - base, d_.data_ptr() - base, n, c) for a, b, g_, d_, n, c in items)
        if self._pg_table is None or self._pg_table[0] != key:
            table, max_c = ops.pgrad_table(items, device, base)
            self._pg_table = (key, table, max_c)
        ops.norm_param_grads_batch(self._pg_table[1], len(items), self._pg_table[2], base)

    def _small_norm_bwd(self, P, G, ctx: UNetCtx, src: Optional[Act], n: int, hw: int, c: int):
        """(target layer, ``norm_bwd_full`` argument) when the input-gradient launch that differentiates ``src`` may carry
        the norm backward of the layer that produced it (maps of <= 64 pixels), else None."""
        tgt = self._producer.get(id(src)) if src is not None else None
        if (tgt is None or not self.small_norm or not self.fused_norm or self.deterministic or self.debug is not None
                or hw > self.small_norm_px or c % 32 != 0):
            return None
        rec = ctx.convs.get(tgt)
        if rec is None or rec.drop_mask is not None or rec.out.stats is None or rec.first:
            return None
        parts = self._pgrad_parts(G, tgt, n, hw, c, rec.out.z.device, register=False)
        if parts is not None:      # registered by _norm_bwd_taken only if the launch takes the epilogue: the layer's own norm
            #                        backward registers planes of its own otherwise, and two items of one layer would race
            return tgt, (rec.out, P[f"{tgt}.norm.weight"], parts[0], parts[1], True), parts[2]
        return tgt, (rec.out, P[f"{tgt}.norm.weight"], G[f"{tgt}.norm.weight"], G[f"{tgt}.norm.bias"], False), None

    def _norm_bwd_taken(self, full):
        """the input-gradient launch carried the norm backward of layer ``full[0]``"""
        self._bwd_done.add(full[0])
        if full[2] is not None:
            self._pg_items.append(full[2])

    def _convT_bwd(self, *a, **k):
        self._convT_bwd_impl(*a, **k)
        self._flush_after_dgrad()

    def _convT_bwd_impl(self, P, G, ctx: UNetCtx, rec: _UpRec, du: Tensor, d_in: Tensor, accum: int):
        w = P[f"{rec.prefix}.weight"]
        ci, co = w.shape[0], w.shape[1]
        n, h, w_, _ = rec.src.z.shape
        taps = [(0, 0, dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)]
        self._wgrad_task([du, rec.src.z] + ([rec.src.a] if rec.src.a is not None else []),
                         lambda: self._wgrad([rec.src], du, (4, co, ci), G[f"{rec.prefix}.weight"], "convT", rec.prefix,
                                             grid=(h, w_), in_stride=1, z_stride=2, taps=taps, n_cols=co))
        _, wd = self._operands(f"{rec.prefix}.weight", w, "convT")
        full = self._small_norm_bwd(P, G, ctx, rec.src, n, h * w_, ci)
        got = ops.conv_gemm([Act(du, None, 1.0)], wd, None, grid=(h, w_), in_stride=2,
                            taps=[(dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)], dsts=[d_in], dst_cols=[ci],
                            accum=[accum], norm_bwd_full=full[1] if full else None)
        if got and full:
            self._norm_bwd_taken(full)

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, P: Dict[str, Tensor], G: Dict[str, Tensor], ctx: UNetCtx, dlogits: Optional[Tensor],
                 dfeats: Optional[Tensor], dl_nhwc: Optional[Tensor] = None, head_grads: Optional[tuple] = None):
        """Accumulates parameter gradients into G (float32, reference layouts; every touched tensor is ``+=``).
        dL/dlogits arrives as ``dlogits`` (N, K, H, W) float32 and / or as ``dl_nhwc`` (N, H, W, 32) in the engine's
        element type (``cu_dsnt_head_bwd_nhwc``); both given = their sum."""
        try:
            self._main_stream = torch.cuda.current_stream(ctx.last.z.device) if ctx.last.z.is_cuda else None
            self._backward(P, G, ctx, dlogits, dfeats, dl_nhwc, head_grads)
            self._main_stream = None
        except BaseException:
            self._main_stream = None
            self._on_side = False
            # an aborted step (launch error, OOM, KeyboardInterrupt, an exception from the DDP ready hook) must not leave
            # state behind that a later step would silently consume: join the weight-gradient stream, drop the epilogue
            # sums and the (possibly half-accumulated) first-layer accumulator
            try:
                if self._side is not None:
                    torch.cuda.current_stream(self._side.device).wait_stream(self._side)
                if self._red is not None:
                    torch.cuda.current_stream(self._red.device).wait_stream(self._red)
            finally:
                self._red_done = [None, None]
                self._side_keep.clear()
                self._lagq.clear()
                self._given_sums.clear()
                self._bwd_done.clear()
                self._pg_items = []
                self._dw9_ws = None
                hook = getattr(self.grad_ready_hook, "__self__", None)
                if hook is not None and hasattr(hook, "abort"):
                    hook.abort()
            raise

    def _head_bwd(self, P, G, ctx: UNetCtx, head_grads) -> Tensor:
        """fused head backward: dL/d(mu, Sigma) -> g = dL/d(activation of the last ConvLayer); that layer's norm-backward sums
        and the 1x1 weight gradient come out of the same launch (cu_head_fused_bwd)."""
        hd = ctx.head
        act: Act = hd["act"]
        n = act.z.shape[0]
        dev = act.z.device
        aux, gmu, gsigma, covar = head_grads
        sums = self._arena["bwd"].take(2 * n * 32, dev)
        if self._head_parts is None or self._head_parts.device != dev:
            self._head_parts = torch.empty(1025 * 1024, dtype=torch.float32, device=dev)
        g, slabs = ops.head_fused_bwd(act, hd["w_cls"], hd["w_ch"], hd["k"], aux, gmu.contiguous(), gsigma.contiguous(),
                                      bool(covar), sums, self._head_parts)
        ops.grad_unprep_parts(self._head_parts, slabs, 32, G["output_block.conv.weight"], "conv", accumulate=True)
        self._ready("output_block")
        self._given_sums[ctx.raw_prefix] = sums
        return g

    def _backward(self, P, G, ctx: UNetCtx, dlogits, dfeats, dl_nhwc, head_grads=None):
        dt = self.dtype
        last = ctx.last
        n, h, w_, c_last = last.z.shape
        self._arena["bwd"].begin(last.z.device)
        self._given_sums.clear()
        self._bwd_done.clear()
        self._pg_items = []
        # the previous backward joined the reduction stream into this one: its "buffer read" events are history (and must not
        # be waited for inside a hipGraph capture, which they precede)
        self._red_done = [None, None]
        if ctx.head is not None:
            assert head_grads is not None and dlogits is None and dl_nhwc is None, "fused head: gradients arrive as head_grads"
            g = self._head_bwd(P, G, ctx, head_grads)
        else:
            g = self._output_bwd(P, G, ctx, dlogits, dl_nhwc)
        self._decoder_encoder_bwd(P, G, ctx, g, dfeats)

    def _output_bwd(self, P, G, ctx: UNetCtx, dlogits, dl_nhwc) -> Tensor:
        dt = self.dtype
        last = ctx.last
        n, h, w_, c_last = last.z.shape
        # ---- 1x1 output conv
        if dl_nhwc is not None:
            assert dl_nhwc.dtype == dt and dl_nhwc.shape == (n, h, w_, 32)
            dl = dl_nhwc if dlogits is None else dl_nhwc + ops.nchw_f32_to_nhwc(dlogits.contiguous(), dt, cp=32)
        else:
            dl = ops.nchw_f32_to_nhwc(dlogits.contiguous(), dt, cp=32)
        w = P["output_block.conv.weight"]
        with self._wgrad_stream(dl, last.z):
            self._wgrad([last], dl, (1, 32, c_last), G["output_block.conv.weight"], "conv", "output_block", grid=(h, w_),
                        in_stride=1, z_stride=1, taps=[(0, 0, 0, 0, 0)], n_cols=32, alg_cols=self.num_classes)
        _, wd = self._operands("output_block.conv.weight", w, "conv", cop=32)
        g = torch.empty_like(last.z)
        ops.conv_gemm([Act(dl, None, 1.0)], wd, None, grid=(h, w_), in_stride=1, taps=[(0, 0, 0)], dsts=[g],
                      dst_cols=[c_last], alg_cin=self.num_classes)
        return g

    def _decoder_encoder_bwd(self, P, G, ctx: UNetCtx, g: Tensor, dfeats):
        dt = self.dtype
        # ---- decoder
        d_enc: List[Optional[Tensor]] = [None] * len(ctx.enc)
        n_up = ctx.n_up
        d_bott = None
        for i in reversed(range(n_up)):
            up = ctx.ups[i]
            pre = f"upsamples.{i}.conv_block"
            c1 = ctx.convs[f"{pre}.conv1"]
            g_c1 = torch.empty_like(c1.out.z)
            self._conv_layer_bwd(P, G, ctx, f"{pre}.conv2", g, [(g_c1, 0)])
            k = n_up - 1 - i                      # encoder level of this skip
            du = torch.empty_like(up.u.z)
            d_enc[k] = torch.empty_like(ctx.enc[k].z)
            self._conv_layer_bwd(P, G, ctx, f"{pre}.conv1", g_c1, [(du, 0), (d_enc[k], 0)])
            del g_c1
            if i == 0:
                if dfeats is not None:
                    ops.pending_wait()       # the skew head's backward may have produced dfeats on its own stream
                    if callable(dfeats):     # ... and then hands it over lazily (unet2._UNetFn.backward)
                        dfeats = dfeats()
                    d_bott = ops.nchw_f32_to_nhwc(dfeats.contiguous(), dt)
                accum = 1 if d_bott is not None else 0
                d_in = d_bott if d_bott is not None else torch.empty_like(up.src.z)
            else:
                accum = 0
                d_in = torch.empty_like(up.src.z)
            self._convT_bwd(P, G, ctx, up, du, d_in, accum)
            del du
            g = d_in
        # ---- bottleneck + encoder
        st = self.strides
        blocks = [("bottleneck", len(ctx.enc) - 1)] + [(f"downsamples.{i}", i) for i in reversed(range(len(st) - 2))]
        for prefix, k_in in blocks:
            c1 = ctx.convs[f"{prefix}.conv1"]
            g_c1 = torch.empty_like(c1.out.z)
            self._conv_layer_bwd(P, G, ctx, f"{prefix}.conv2", g, [(g_c1, 0)])
            self._conv_layer_bwd(P, G, ctx, f"{prefix}.conv1", g_c1, [(d_enc[k_in], 1)])
            g = d_enc[k_in]
        c1 = ctx.convs["input_block.conv1"]
        g_c1 = torch.empty_like(c1.out.z)
        self._conv_layer_bwd(P, G, ctx, "input_block.conv2", g, [(g_c1, 0)])
        self._conv_layer_bwd(P, G, ctx, "input_block.conv1", g_c1, None)
        self._pgrad_finish(g.device)
        self._join_wgrad(g.device)


class ConfidenceEngine:
    """ConfidenceNet (reference models/nnUnet/unet2.py:14-34): 3 x (conv3x3 + ReLU) -> flatten -> Linear."""

    def __init__(self, dtype: torch.dtype = torch.bfloat16):
        self.dtype = dtype
        self._opcache: Dict[str, Tuple] = {}
        self._parts: Optional[Tensor] = None        # partial-tile scratch of the three weight gradients
        self.deterministic = os.environ.get("CONTOUR_DETERMINISTIC", "0") == "1"      # as UNetEngine.deterministic

    def _operands(self, name, w):
        key = (w.data_ptr(), w._version, ops.PARAM_EPOCH[0])
        hit = self._opcache.get(name)
        if hit is not None and hit[0] == key and hit[1].dtype == self.dtype:
            return hit[1], hit[2]
        wf, wd = ops.weight_prep(w.detach(), "conv", self.dtype)
        self._opcache[name] = (key, wf, wd)
        return wf, wd

    def forward(self, P: Dict[str, Tensor], feats: Tensor):
        """feats (N, 480, h, w) float32 NCHW -> (N, out) float32; returns (out, ctx)."""
        n, c, h, w_ = feats.shape
        x = Act(ops.nchw_f32_to_nhwc(feats.contiguous(), self.dtype), None, 1.0)
        acts = [x]
        for i in (0, 2, 4):
            w = P[f"model.{i}.weight"]
            wf, _ = self._operands(f"model.{i}.weight", w)
            z = torch.empty((n, h, w_, w.shape[0]), dtype=self.dtype, device=feats.device)
            ops.conv_gemm([acts[-1]], wf, P[f"model.{i}.bias"], grid=(h, w_), in_stride=1, taps=TAPS3, dsts=[z],
                          dst_cols=[w.shape[0]])
            acts.append(Act(z, None, 0.0))          # ReLU applied by the consumer
        flat = ops.act_to_nchw_f32(acts[-1]).view(n, -1)      # nn.Flatten of the NCHW tensor
        out = ops.linear_fwd(flat, P["model.7.weight"], P["model.7.bias"])
        return out, (acts, flat)

    def backward(self, P, G, ctx, gout: Tensor, need_input_grad: bool = True):
        acts, flat = ctx
        n, h, w_, c3 = acts[-1].z.shape
        gflat = ops.linear_bwd(flat, P["model.7.weight"], gout.contiguous(), G["model.7.weight"], G["model.7.bias"])
        g = ops.nchw_f32_to_nhwc(gflat.view(n, c3, h, w_), self.dtype)
        for li, i in reversed(list(enumerate((0, 2, 4)))):
            src, out = acts[li], acts[li + 1]
            w = P[f"model.{i}.weight"]
            co, ci = w.shape[0], w.shape[1]
            ops.act_bwd(g, out.z, 0.0, G[f"model.{i}.bias"], deterministic=self.deterministic)
            if self._parts is None or self._parts.device != g.device:
                self._parts = torch.empty(4 << 20, dtype=torch.float32, device=g.device)
            slabs = ops.conv_wgrad([src], g, self._parts, grid=(h, w_), in_stride=1, z_stride=1, taps=TAPS3_W, n_cols=co,
                                    parts=True)
            ops.grad_unprep_parts(self._parts, slabs, co, G[f"model.{i}.weight"], "conv", accumulate=True)
            if li == 0 and not need_input_grad:
                return None
            _, wd = self._operands(f"model.{i}.weight", w)
            gin = torch.empty_like(src.z)
            ops.conv_gemm([Act(g, None, 1.0)], wd, None, grid=(h, w_), in_stride=1, taps=TAPS3_D, dsts=[gin],
                          dst_cols=[ci])
            g = gin
        return ops.nhwc_to_nchw_f32(g)
