"""``contour_uncertainty.task.uncertainty.UncertaintyTask`` (reference task/uncertainty.py:27-150): boilerplate common
to all uncertainty methods.  Only what the DSNT train/val/predict path uses is kept; MC-dropout patching, ensembling
figure upload is a host-side feature outside the accelerated path.  ``t_e > 1`` keeps
Dropout2d active at predict time (MC dropout, reference utils/mcdropout.py:89-137) and ``ensemble_ckpt`` loads one
network per checkpoint (reference uncertainty.py:55-70)."""
from __future__ import annotations

from typing import Any, Dict, List, Union

import numpy as np
import torch
from torch import Tensor

from contour_uncertainty._compat import SharedStepsTask, prefix
from contour_uncertainty.utils.metrics import Dice


class UncertaintyTask(SharedStepsTask):
    def __init__(self, t_a: int = 1, t_e: int = 1, train_ensemble: bool = False, ensemble_ckpt=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        labels = getattr(self.hparams.data_params, "labels", None)
        self.dice = Dice(labels=labels)
        self.model = self.configure_model()
        self.is_val_step = False
        if ensemble_ckpt is not None:
            from pathlib import Path
            from torch import nn
            from contour_uncertainty._compat import resolve_model_checkpoint_path
            self.ensembling = True
            if isinstance(ensemble_ckpt, (list, tuple)):
                files = list(ensemble_ckpt)
            elif Path(str(ensemble_ckpt)).is_dir():
                files = sorted(Path(str(ensemble_ckpt)).glob("*.ckpt"))
            else:
                raise ValueError("ENSEMBLE not valid")
            self.model = nn.ModuleList(
                [self.load_from_checkpoint(resolve_model_checkpoint_path(w), ensemble_ckpt=None).model for w in files])
            self.hparams.t_e = len(self.model)
        else:
            self.ensembling = False
            if self.hparams.t_e > 1:
                # keep dropout at test time (reference: patch_module swaps nn.Dropout2d for an always-on layer)
                if getattr(self.model, "drop_block", False):
                    self.model.mc_dropout = True
                else:
                    import warnings
                    warnings.warn("No layer was modified by patch_module!", UserWarning)

    def on_fit_start(self) -> None:
        """``train_ensemble=True``: every ensemble member trains on its own random 90 % of the training set (reference
        task/uncertainty.py:76-80 swaps the datamodule's TRAIN dataset for a ``torch.utils.data.Subset``)."""
        if not self.hparams.get("train_ensemble"):
            return
        import random
        from torch.utils.data import Subset as TorchSubset
        dm = getattr(getattr(self, "trainer", None), "datamodule", None)
        sets = getattr(dm, "_dataset", None)
        if sets is None:
            raise RuntimeError("train_ensemble=True needs trainer.datamodule._dataset (vital's VitalDataModule layout) to "
                               "draw the member's 90 % training subset from")
        key = next((k for k in sets if str(getattr(k, "value", k)).lower() == "train"), None)
        if key is None:
            raise RuntimeError("train_ensemble=True: the datamodule has no TRAIN subset")
        full = sets[key]
        keep = random.sample(range(len(full)), int(0.9 * len(full)))
        sets[key] = TorchSubset(full, keep)

    def on_fit_end(self) -> None:
        logger = getattr(getattr(self, "trainer", None), "logger", None)
        if logger is not None and hasattr(logger, "log_hyperparams"):
            logger.log_hyperparams({"train_complete": True})          # reference task/uncertainty.py:82-83

    def forward(self, *args, **kwargs):  # noqa: D102
        return self.model(*args, **kwargs)

    def validation_step(self, *args, **kwargs) -> Dict[str, Tensor]:  # noqa: D102
        self.is_val_step = True
        result = prefix(self._shared_step(*args, **kwargs), "val/")
        self.is_val_step = False
        self.log_dict(result, **(self.hparams.get("val_log_kwargs") or {}))
        self.log("val_loss", result["val/loss"], on_step=True, on_epoch=True, prog_bar=True, logger=False)
        return result

    def predict_step(self, batch: Any, batch_idx: int, dataloader_idx: int = 0):
        raise NotImplementedError

    @staticmethod
    def sample_entropy(samples):
        """reference uncertainty.py:107-133: entropy of the mean sample map (binary case when C == 1)."""
        import scipy.stats
        samples = torch.as_tensor(samples)
        if samples.ndim == 5:
            samples = samples.reshape(-1, *samples.shape[2:])
        y_hat = samples.mean(0)
        if samples.shape[1] == 1:
            y_hat = torch.cat([y_hat, 1 - y_hat], dim=0)
            base = 2
        else:
            base = samples.shape[1]
        umap = scipy.stats.entropy(y_hat.cpu().numpy(), axis=0, base=base)
        umap[~np.isfinite(umap)] = 0
        return umap
