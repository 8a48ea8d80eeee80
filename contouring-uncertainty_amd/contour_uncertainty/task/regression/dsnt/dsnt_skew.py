"""``contour_uncertainty.task.regression.dsnt.dsnt_skew.DSNTSkew`` on the MI355X kernels.

Reference: task/regression/dsnt/dsnt_skew.py:18-199 (Hydra target of config/task/dsnt-skew*.yaml).  The skew head is a
sibling module (``self.skew_block``), so checkpoint keys are ``model.*`` and ``skew_block.model.*`` as in the reference.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from contour_uncertainty._compat import ContourTags, Tags, fused_optimizer_cfg, instantiate
from contour_uncertainty.task.regression.aleatoric_skew import SkewUncertaintyTask
from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
from cu_hip.head import dsnt_moments, dsnt_nll


class DSNTSkew(SkewUncertaintyTask):
    """Reference: https://github.com/anibali/dsntnn"""

    def __init__(self, covar: bool = True, mse_weight: float = 1, log_penalty_weight: float = 1,
                 freeze_seg: bool = False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        if self.ensembling:
            # the reference fails here too: nn.ModuleList has no confidence_net (dsnt_skew.py:34 after uncertainty.py:58)
            raise NotImplementedError("ensemble_ckpt is not usable with dsnt-skew (one skew head, many networks)")
        self.skew_block = self.model.confidence_net(len(self.skew_indices) * 2)
        if hasattr(self.model, "engine"):
            self.skew_block.set_compute_dtype(self.model.engine.dtype)
        if freeze_seg:
            self.freeze_layers()

    def freeze_layers(self):
        """Freezes the U-Net for fine-tuning of the skew head (reference dsnt_skew.py:39-44)."""
        for _, p in self.model.named_parameters():
            p.requires_grad = False

    def configure_model(self) -> nn.Module:
        in_shape = self.hparams.data_params.in_shape
        out_shape = self.hparams.data_params.out_shape
        return instantiate(self.hparams.model, input_shape=in_shape,
                           output_shape=(out_shape[0], in_shape[0], in_shape[1]), bottleneck_out=True)

    configure_optimizers = DSNTAleatoric.configure_optimizers
    _val_dice = DSNTAleatoric._val_dice

    def _alpha(self, heatmaps: Tensor, features: Tensor, side: bool = False) -> Tensor:
        """(N, K*, 2) head output scattered into zeros (N, K, 2) at skew_indices (reference dsnt_skew.py:68-71).
        ``side``: the head may run on its own stream beside the U-Net (``ConfidenceNet.forward(side=True)``); the result then
        goes straight to ``dsnt_nll``, which waits for it."""
        n, k = heatmaps.shape[0], heatmaps.shape[1]
        if side and hasattr(self.skew_block, "side_enabled"):
            a = self.skew_block(features, side=True).view(n, len(self.skew_indices), 2)
        else:
            a = self.skew_block(features).view(n, len(self.skew_indices), 2)
        if len(self.skew_indices) == k:
            return a
        from cu_hip import ops as _ops
        _ops.pending_wait()          # the scatter below reads the head's output on the current stream
        alpha = torch.zeros(n, k, 2, device=a.device, dtype=a.dtype)
        alpha[:, self.skew_indices, :] = a
        return alpha

    def _shared_step(self, batch: Dict[str, Tensor], batch_idx: int) -> Dict[str, Tensor]:  # noqa: D102
        x, y = batch[Tags.img], batch[ContourTags.contour]
        # (inside fused_head() the heat maps may be a placeholder that only dsnt_nll reads: cu_hip/head_fused.hip)
        with (self.model.fused_head() if hasattr(self.model, "fused_head") else contextlib.nullcontext()):
            heatmaps, features = self.model(x)
        alpha = self._alpha(heatmaps, features, side=True)
        logs, pixel_coords, _ = dsnt_nll(heatmaps, y, alpha, self.hparams.covar)
        if self.is_val_step and Tags.gt in batch:
            self._val_dice(logs, batch, x, pixel_coords)
        return logs

    def predict_on_batch(self, img, model):
        """-> mu (N,K,2), Sigma (N,K,2,2), alpha (N,K,2) with alpha_y negated (reference dsnt_skew.py:153-176, :164)"""
        with torch.no_grad():
            heatmaps, features = model(img)
            alpha = self._alpha(heatmaps, features).clone()
            alpha[..., 1] = -alpha[..., 1]
            mu, sigma = dsnt_moments(heatmaps, self.hparams.covar)
        return mu, sigma, alpha

    def predict(self, img, scale=False) -> Tuple:  # noqa: D102
        S, cov, alpha = [], [], []
        self.hparams.t_e = len(self.model) if self.ensembling else self.hparams.t_e
        for i in range(self.hparams.t_e):
            model = self.model[i] if self.ensembling else self.model
            m, s, a = self.predict_on_batch(img, model)
            S.append(m)
            cov.append(s)
            alpha.append(a)
        S = torch.stack(S).swapaxes(1, 0)
        cov = torch.stack(cov).swapaxes(1, 0)
        alpha = torch.stack(alpha).swapaxes(1, 0)
        return S.cpu().detach(), cov.cpu().detach(), alpha.cpu().detach()
