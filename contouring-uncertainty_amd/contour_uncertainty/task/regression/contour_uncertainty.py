"""``ContourUncertaintyTask`` (reference task/regression/contour_uncertainty.py:23-130): sample -> mask conversion and
the predict_step post-processing.  Host NumPy work that SURVEY.md 8(f) ranks as the next thing to move on-device."""
from __future__ import annotations

from typing import Any, Tuple

import numpy as np
from numpy import linalg as LA

from contour_uncertainty.task.uncertainty import UncertaintyTask
from contour_uncertainty.utils.contour import contour_to_mask


class ContourUncertaintyTask(UncertaintyTask):
    contour_to_mask_fn = staticmethod(contour_to_mask)
    umap_fn = None
    skew_umap_fn = None

    def convert_to_mask(self, mu: np.ndarray, image_shape: Tuple, contour_samples=None):
        """reference contour_uncertainty.py:26-57 (soft_mask branch unused by the dsnt tasks)."""
        n = image_shape[0]
        labels = self.hparams.data_params.labels
        pred = np.array([self.contour_to_mask_fn(mu[i], image_shape[-2:], labels) for i in range(n)])
        pred_samples = None
        if contour_samples is not None:
            t_e, t_a = contour_samples.shape[1], contour_samples.shape[2]
            pred_samples = np.array([
                self.contour_to_mask_fn(contour_samples[i, j, k], image_shape[-2:], labels, apply_argmax=False)
                for i in range(n) for j in range(t_e) for k in range(t_a)
            ]).reshape(n, t_e, t_a, *image_shape[1:])
        return pred, pred_samples

    def _bind_datamodule_fns(self):
        dm = getattr(getattr(self, "trainer", None), "datamodule", None)
        if dm is not None:
            self.umap_fn = staticmethod(dm.umap_fn)
            self.contour_to_mask_fn = staticmethod(dm.contour_to_mask_fn)
            self.skew_umap_fn = staticmethod(dm.skew_umap_fn)

    def on_predict_start(self):
        self._bind_datamodule_fns()

    def on_fit_start(self):
        self._bind_datamodule_fns()

    def predict_step(self, batch: Any, batch_idx: int, dataloader_idx: int = 0):
        """reference contour_uncertainty.py:71-130: point / instant uncertainty summaries of a BatchResult."""
        res = self._predict_step(batch)
        n = res.img.shape[0]
        if res.pred_samples is not None:
            res.entropy_map = np.array([self.sample_entropy(res.pred_samples[i].astype(float)) for i in range(n)])
            res.pred_samples = res.pred_samples.squeeze(3)
        cov_det = LA.det(res.cov) ** 0.25
        cov_eigval = np.sqrt(LA.eig(res.cov)[0])
        res.point_uncertainty = {"cov_xx": res.cov[:, :, 0, 0] ** 0.5, "cov_yy": res.cov[:, :, 1, 1] ** 0.5,
                                 "cov_det": cov_det, "cov_eigval_sum": cov_eigval.sum(-1)}
        if res.post_cov is not None:
            pe = np.sqrt(LA.eig(res.post_cov)[0])
            res.point_uncertainty.update({"post_cov_xx": res.post_cov[:, :, 0, 0] ** 0.5,
                                          "post_cov_yy": res.post_cov[:, :, 1, 1] ** 0.5,
                                          "post_cov_det": LA.det(res.post_cov) ** 0.25,
                                          "post_cov_eigval_sum": pe.sum(-1)})
        res.instant_uncertainty = {"cov_det_mean": np.mean(cov_det, axis=-1),
                                   "cov_eigenvalue_mean": np.mean(cov_eigval, axis=(-1, -2))}
        return res

    def _predict_step(self, batch: Any):
        raise NotImplementedError
