"""DSNT head + NLL as one autograd node on the HIP kernels.

Replaces, for the tasks' ``_shared_step`` (reference task/regression/dsnt/dsnt_al.py:52-74, dsnt_skew.py:73-104):
``flat_softmax`` -> ``dsnt`` -> ``normalized_to_pixel_coordinates`` -> ``get_cov_matrix`` -> Gaussian / skew-normal NLL
and the logged means.  Forward runs ``cu_dsnt_head_fwd`` + ``cu_nll_fwd_bwd`` (which also yields the analytic
gradients w.r.t. mu / Sigma / alpha); backward is one ``cu_dsnt_head_bwd`` streaming pass.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops

Tensor = torch.Tensor

LOG_KEYS = ("loss", "distance_loss", "loss_term1", "loss_term2", "loss_term3", "alpha_norm")


class GradSlot:
    """Hand-over of dL/dlogits in the layout the producer's backward reads.  ``UNet.forward`` hangs one on the logits it
    returns (``logits._cu_grad_slot``); the head's backward then writes the gradient ONCE, as NHWC in the engine's
    element type (``cu_dsnt_head_bwd_nhwc``), puts it here and returns a stride-0 zero tensor to autograd; the UNet's
    backward takes it (and adds whatever else autograd accumulated into its incoming gradient).  Logits that did not
    come straight from such a UNet carry no slot and get the ordinary NCHW float32 gradient.

    Several heads on the same logits (two loss terms, a retained graph walked twice) ADD into the slot (an
    overwrite lost every gradient but the last).  What the hand-over cannot serve is a reader of the logits' own
    gradient -- ``torch.autograd.grad(loss, logits)``, ``logits.retain_grad()``, a tensor hook on the logits: those see the
    zero stand-in.  ``dsnt_nll(..., dense_grad=True)`` (or ``slot.enabled = False``) selects the dense NCHW float32
    gradient for such uses."""
    __slots__ = ("dtype", "dl", "enabled", "fused", "head", "head_grads", "feats_event", "feats_grad")

    def __init__(self, dtype):
        self.dtype = dtype
        self.dl: Optional[Tensor] = None
        self.enabled = True
        # fused head (head_fused.hip; ``UNet.fused_head()``): ``fused`` = asked for, ``head`` = the engine's handle when the
        # forward took it (the logits are then a placeholder), ``head_grads`` = (aux, dL/dmu, dL/dSigma3, covar) on the way back
        self.fused = False
        self.head: Optional[dict] = None
        self.head_grads: Optional[tuple] = None
        self.feats_event = None      # recorded when the bottleneck features of the same forward exist
        # dL/d(bottleneck features) left by a consumer that ran its backward on a stream of its own (ConfidenceNet side mode):
        # readable only behind ops.pending_wait(); autograd carries a stride-0 zero stand-in meanwhile
        self.feats_grad: Optional[Tensor] = None

    def take_feats(self) -> Optional[Tensor]:
        g, self.feats_grad = self.feats_grad, None
        return g

    def put_head(self, aux: Tensor, gmu: Tensor, gsigma: Tensor, covar: bool):
        if not covar:
            gsigma = gsigma.clone()
            gsigma[..., 2] = 0
        if self.head_grads is None:
            self.head_grads = (aux, gmu, gsigma, True)
        else:       # several heads on the same logits: dL/dlogits is linear in (dL/dmu, dL/dSigma)
            a0, m0, s0, _ = self.head_grads
            self.head_grads = (a0, m0 + gmu, s0 + gsigma, True)

    def take_head(self) -> Optional[tuple]:
        hg, self.head_grads = self.head_grads, None
        return hg

    def put(self, dl: Tensor):
        self.dl = dl if self.dl is None else self.dl + dl

    def take(self) -> Optional[Tensor]:
        dl, self.dl = self.dl, None
        return dl


class _DsntNllFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, y: Tensor, alpha: Optional[Tensor], covar: bool, w_mse: float, w_log: float,
                slot: Optional[GradSlot] = None):
        fused = slot is not None and slot.head is not None
        ctx.fused = fused
        if fused:
            ctx.slot = slot          # ``logits`` is the placeholder of UNet.fused_head(): never read
        else:
            ctx.slot = slot if (slot is not None and slot.enabled and logits.is_contiguous()) else None
            logits = logits.contiguous()
        need_grad = logits.requires_grad or (alpha is not None and alpha.requires_grad)
        with ops.L.device_guard(y if fused else logits):
            if fused:
                hd = slot.head
                mu, sigma, aux = ops.head_fused_fwd(hd["act"], hd["w_cls"], hd["k"], covar)
                logits = logits.new_empty(0)
            else:
                mu, sigma, aux = ops.dsnt_head_fwd(logits, covar)
            if alpha is not None:
                ops.pending_wait()          # alpha may come from the skew head's own stream (ConfidenceNet side mode)
            al = alpha.contiguous().float() if alpha is not None else None
            logs, gmu, gsigma, galpha = ops.nll_fwd_bwd(mu, sigma, y.contiguous().float(), al, w_mse, w_log, need_grad)
        ctx.covar = covar
        ctx.has_alpha = alpha is not None
        ctx.map_hw = (slot.head["act"].z.shape[1], slot.head["act"].z.shape[2]) if fused else None
        if need_grad:
            ctx.save_for_backward(logits, aux, gmu, gsigma, galpha if galpha is not None else torch.empty(0))
        ctx.mark_non_differentiable(logs, mu, sigma)
        return logs[0].clone(), logs, mu, sigma

    @staticmethod
    def backward(ctx, gloss, _glogs, _gmu, _gsigma):
        logits, aux, gmu, gsigma, galpha = ctx.saved_tensors
        scale = gloss.reshape(1)
        if ctx.fused:
            # dL/d(mu, Sigma) go to the UNet's backward, which runs cu_head_fused_bwd (no dL/dlogits anywhere)
            ctx.slot.put_head(aux, (gmu * scale).contiguous(), (gsigma * scale).contiguous(), ctx.covar)
            shape = (gmu.shape[0], gmu.shape[1]) + tuple(ctx.map_hw)
            dl = ops.zero_placeholder(shape, torch.float32, gmu.device)
            return dl, None, ((galpha * scale) if ctx.has_alpha else None), None, None, None, None
        with ops.L.device_guard(logits):
            if ctx.slot is not None:
                ctx.slot.put(ops.dsnt_head_bwd_nhwc(logits, aux, (gmu * scale).contiguous(), (gsigma * scale).contiguous(),
                                                    ctx.covar, ctx.slot.dtype))
                dl = ops.zero_placeholder(logits.shape, logits.dtype, logits.device)
            else:
                dl = ops.dsnt_head_bwd(logits, aux, (gmu * scale).contiguous(), (gsigma * scale).contiguous(), ctx.covar)
        dalpha = (galpha * scale) if ctx.has_alpha else None
        return dl, None, dalpha, None, None, None, None


def dsnt_nll(logits: Tensor, y: Tensor, alpha: Optional[Tensor] = None, covar: bool = True, mse_weight: float = 1.0,
             log_penalty_weight: float = 1.0, dense_grad: bool = False):
    """logits (N,K,H,W) f32, y (N,K,2) pixel (x,y), alpha (N,K,2) or None ->
    (logs dict of 0-dim tensors with a differentiable ``loss``, mu (N,K,2), Sigma (N,K,2,2)).
    ``dense_grad``: give autograd the real NCHW float32 dL/dlogits instead of the :class:`GradSlot` hand-over."""
    slot = getattr(logits, "_cu_grad_slot", None)
    if slot is not None and slot.head is not None:
        if dense_grad:
            raise ValueError("dense_grad=True needs real logits: call the UNet outside `with model.fused_head():`")
    elif dense_grad:
        slot = None
    loss, logs, mu, sigma3 = _DsntNllFn.apply(logits, y, alpha, bool(covar), float(mse_weight),
                                              float(log_penalty_weight), slot)
    out: Dict[str, Tensor] = {"loss": loss, "distance_loss": logs[1], "loss_term1": logs[2], "loss_term2": logs[3]}
    if alpha is not None:
        out["loss_term3"] = logs[4]
        out["alpha_norm"] = logs[5]
    return out, mu, sigma_matrix(sigma3)


def sigma_matrix(sigma3: Tensor) -> Tensor:
    """{xx, yy, xy} -> (..., 2, 2) exactly as get_cov_matrix (reference aleatoric.py:138-144)."""
    xx, yy, xy = sigma3[..., 0], sigma3[..., 1], sigma3[..., 2]
    return torch.stack([xx, xy, xy, yy], -1).unflatten(-1, (2, 2))       # (one launch instead of three)


@torch.no_grad()
def dsnt_moments(logits: Tensor, covar: bool = True):
    """predict-time head: pixel mu (N,K,2), Sigma (N,K,2,2)  (reference dsnt_al.py:118-131)."""
    mu, sigma3, _ = ops.dsnt_head_fwd(logits.contiguous(), covar)
    return mu, sigma_matrix(sigma3)
