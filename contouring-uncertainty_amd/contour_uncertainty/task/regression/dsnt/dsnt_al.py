"""``contour_uncertainty.task.regression.dsnt.dsnt_al.DSNTAleatoric`` on the MI355X kernels.

Same constructor keys, step-dict keys and predict outputs as the reference class
(reference task/regression/dsnt/dsnt_al.py:17-151; Hydra target of config/task/dsnt-al.yaml:1).
``_shared_step`` = U-Net (cu_hip.engine) -> fused DSNT head + Gaussian NLL (cu_hip.head); the loss is
``mean(t1) + mean(t2)``, the exact value of the reference's (NK,1,NK) broadcast mean.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from contour_uncertainty._compat import ContourTags, Tags, fused_optimizer_cfg, instantiate
from contour_uncertainty.task.regression.aleatoric import AleatoricUncertaintyTask
from cu_hip.head import dsnt_moments, dsnt_nll


class DSNTAleatoric(AleatoricUncertaintyTask):
    """Reference: https://github.com/anibali/dsntnn"""

    def __init__(self, covar: bool = True, mse_weight: float = 1, log_penalty_weight: float = 1, iterations: int = 1,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()

    def configure_model(self) -> nn.Module:
        in_shape = self.hparams.data_params.in_shape
        out_shape = self.hparams.data_params.out_shape
        # reference dsnt_al.py:41-43 passes (K, in_shape[0], in_shape[1]); only output_shape[0] is read (unet2.py:102)
        return instantiate(self.hparams.model, input_shape=in_shape,
                           output_shape=(out_shape[0], in_shape[0], in_shape[1]))

    def configure_optimizers(self, params=None):
        """reference vital/vital/system.py:82-115; torch.optim.Adam is served by the fused HIP Adam (same semantics)."""
        if params is None:
            params = self.parameters()
        cfg = self.hparams.optim
        if cfg.get("optimizer"):
            return super().configure_optimizers(params)
        return {"optimizer": instantiate(fused_optimizer_cfg(cfg), params=params)}

    def _val_dice(self, logs, batch, x, pixel_coords):
        gt = batch[Tags.gt]
        pred = np.array([self.contour_to_mask_fn(pixel_coords[i].detach().cpu().squeeze().numpy(),
                                                 (gt.shape[1], gt.shape[2]), labels=self.hparams.data_params.labels,
                                                 reconstruction_type="linear") for i in range(len(x))])
        logs["dice"] = self.dice(pred, gt.cpu().numpy())

    def _shared_step(self, batch: Dict[str, Tensor], batch_idx: int) -> Dict[str, Tensor]:  # noqa: D102
        x, y = batch[Tags.img], batch[ContourTags.contour]
        # (inside fused_head() the heat maps may be a placeholder that only dsnt_nll reads: cu_hip/head_fused.hip)
        with (self.model.fused_head() if hasattr(self.model, "fused_head") else contextlib.nullcontext()):
            heatmaps = self.model(x)
        logs, pixel_coords, _ = dsnt_nll(heatmaps, y, None, self.hparams.covar, self.hparams.mse_weight,
                                         self.hparams.log_penalty_weight)
        if self.is_val_step and Tags.gt in batch:
            self._val_dice(logs, batch, x, pixel_coords)
        return logs

    def predict_on_batch(self, img, model):
        """-> pixel_coords (N, K, 2), pixel_sigma (N, K, 2, 2)  (reference dsnt_al.py:118-131)"""
        with torch.no_grad():
            return dsnt_moments(model(img), self.hparams.covar)

    def predict(self, img, scale=False) -> Tuple:  # noqa: D102
        S, cov = [], []
        self.hparams.t_e = len(self.model) if self.ensembling else self.hparams.t_e
        for i in range(self.hparams.t_e):
            model = self.model[i] if self.ensembling else self.model
            pixel_coords, pixel_sigma = self.predict_on_batch(img, model)
            S.append(pixel_coords)
            cov.append(pixel_sigma)
        S = torch.stack(S).swapaxes(1, 0)          # (N, T_e, K, 2)
        cov = torch.stack(cov).swapaxes(1, 0)
        return S.cpu().detach(), cov.cpu().detach()
