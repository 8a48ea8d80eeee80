"""Oracle (test infrastructure): one whole DSNT training step on PyTorch-CPU, op for op the reference.

Composition follows
  * ``DSNTAleatoric._shared_step``  reference contour_uncertainty/task/regression/dsnt/dsnt_al.py:45-74
  * ``DSNTSkew._shared_step``       reference contour_uncertainty/task/regression/dsnt/dsnt_skew.py:61-104
  * optimiser                       reference vital/vital/system.py:82-115 + vital/vital/config/task/optim/adam.yaml:1-4
                                    (torch.optim.Adam(lr=1e-3, weight_decay=1e-3): L2 folded into the gradient)
This is also the ``cpu_baseline`` that bench.py times on the GPU box's host cores (kind "port").
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import head as H
from . import unet as U

Tensor = torch.Tensor


class OracleTask:
    """Minimal stand-in for the LightningModule: owns leaf tensors, runs fwd/bwd/Adam on CPU."""

    def __init__(self, spec: U.UNetSpec, task: str = "dsnt-skew", covar: bool = True,
                 skew_indices: Optional[Sequence[int]] = None, seed: int = 0, lr: float = 1e-3,
                 weight_decay: float = 1e-3, mse_weight: float = 1.0, log_penalty_weight: float = 1.0,
                 state: Optional[Dict[str, Tensor]] = None, skew_state: Optional[Dict[str, Tensor]] = None):
        assert task in ("dsnt-al", "dsnt-al2", "dsnt-skew")
        self.spec, self.task, self.covar = spec, task, covar
        self.mse_weight, self.log_penalty_weight = mse_weight, log_penalty_weight
        self.skew = task == "dsnt-skew"
        self.skew_indices = list(range(spec.num_classes)) if skew_indices is None else list(skew_indices)
        g = torch.Generator().manual_seed(seed)
        sd = state if state is not None else U.init_unet_state(spec, g)
        self.sd = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        self.skew_sd: Dict[str, Tensor] = {}
        if self.skew:
            ssd = skew_state if skew_state is not None else U.init_confidence_state(2 * len(self.skew_indices), g)
            self.skew_sd = {k: v.detach().clone().requires_grad_(True) for k, v in ssd.items()}
        # reference: self.parameters() = model.* then skew_block.* (dsnt_skew.py:34 registers skew_block after model)
        params = list(self.sd.values()) + list(self.skew_sd.values())
        self.opt = torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)

    def forward_loss(self, img: Tensor, contour: Tensor, masks=None, round_bf16: bool = False) -> Dict[str, Tensor]:
        """masks / round_bf16: test aids, see oracle.unet._LeakyGivenMask / _RoundBf16."""
        if self.skew:
            logits, feats = U.unet_forward(self.sd, img, self.spec, bottleneck_out=True, masks=masks,
                                           round_bf16=round_bf16)
            a = U.confidence_forward(self.skew_sd, feats)
            alpha = H.scatter_alpha(a, self.spec.num_classes, self.skew_indices)
            return H.dsnt_skew_loss(logits, alpha, contour, self.covar)
        logits = U.unet_forward(self.sd, img, self.spec, masks=masks, round_bf16=round_bf16)
        return H.dsnt_al_loss(logits, contour, self.covar, self.mse_weight, self.log_penalty_weight)

    def train_step(self, img: Tensor, contour: Tensor, masks=None) -> Dict[str, float]:
        self.opt.zero_grad(set_to_none=True)
        logs = self.forward_loss(img, contour, masks=masks)
        logs["loss"].backward()
        self.opt.step()
        return {k: float(v.detach()) for k, v in logs.items()}

    @torch.no_grad()
    def predict_on_batch(self, img: Tensor):
        """dsnt_al.py:118-131 / dsnt_skew.py:153-176 (alpha_y negated at predict time only, :164)."""
        if self.skew:
            logits, feats = U.unet_forward(self.sd, img, self.spec, bottleneck_out=True)
            a = U.confidence_forward(self.skew_sd, feats)
            alpha = H.scatter_alpha(a, self.spec.num_classes, self.skew_indices)
            alpha[..., 1] = -alpha[..., 1]
            mu, sigma = H.head_moments(logits, self.covar)
            return mu, sigma, alpha
        logits = U.unet_forward(self.sd, img, self.spec)
        return H.head_moments(logits, self.covar)


def synthetic_batch(n: int, size: int, k: int = 21, seed: int = 1234, device="cpu"):
    """img ~ U[0,1) (N,1,S,S); contour = jittered ellipse arc in pixel (x,y)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, 1, size, size, generator=g)
    t = torch.linspace(0.0, torch.pi, k)[None]
    c = size / 2.0
    rx = (0.16 + 0.19 * torch.rand(n, 1, generator=g)) * size
    ry = (0.16 + 0.19 * torch.rand(n, 1, generator=g)) * size
    x = c + rx * torch.cos(t) + torch.randn(n, k, generator=g) * (size / 128.0)
    y = c - ry * torch.sin(t) + 0.15 * size + torch.randn(n, k, generator=g) * (size / 128.0)
    contour = torch.stack([x, y], dim=-1).clamp(1.0, size - 2.0)
    return img.to(device), contour.to(device)
