"""``ContourUncertaintyTask`` (reference task/regression/contour_uncertainty.py:23-130): sample -> mask conversion and
the predict_step post-processing.  When the bound ``contour_to_mask_fn`` offers ``batch`` (``USContourToMask`` does),
all contours of the step are rasterised by one ``cu_contour_masks`` launch and the sample entropy map comes from
``cu_mask_entropy`` on the bit-packed masks; any other callable is applied per contour, as the
reference does."""
from __future__ import annotations

from typing import Any, Tuple

import numpy as np
import torch
from numpy import linalg as LA

from contour_uncertainty.task.uncertainty import UncertaintyTask
from contour_uncertainty.data.camus.utils import USContourToMask, USSkewUmap, USUMap


class ContourUncertaintyTask(UncertaintyTask):
    contour_to_mask_fn = USContourToMask()          # the CAMUS datamodule's (reference data/camus/datamodule.py:72)
    umap_fn = USUMap()                              # the CAMUS datamodule's (reference data/camus/datamodule.py:73)
    skew_umap_fn = USSkewUmap()                     # the CAMUS datamodule's (reference data/camus/datamodule.py:74)
    _entropy_map = None                             # device-computed entropy of the last convert_to_mask call

    def convert_to_mask(self, mu: np.ndarray, image_shape: Tuple, contour_samples=None):
        """reference contour_uncertainty.py:26-57 (soft_mask branch unused by the dsnt tasks)."""
        n = image_shape[0]
        labels = self.hparams.data_params.labels
        self._entropy_map = None
        fn = self.contour_to_mask_fn
        batch_fn = getattr(getattr(fn, "__func__", fn), "batch", None)      # the datamodule's arrives in a staticmethod
        if batch_fn is not None and len(image_shape) == 4 and image_shape[1] == 1:
            return self._convert_to_mask_device(batch_fn, mu, image_shape, contour_samples, labels)
        pred = np.array([self.contour_to_mask_fn(mu[i], image_shape[-2:], labels) for i in range(n)])
        pred_samples = None
        if contour_samples is not None:
            t_e, t_a = contour_samples.shape[1], contour_samples.shape[2]
            pred_samples = np.array([
                self.contour_to_mask_fn(contour_samples[i, j, k], image_shape[-2:], labels, apply_argmax=False)
                for i in range(n) for j in range(t_e) for k in range(t_a)
            ]).reshape(n, t_e, t_a, *image_shape[1:])
        return pred, pred_samples

    def _convert_to_mask_device(self, batch_fn, mu, image_shape, contour_samples, labels):
        n, (h, w) = image_shape[0], image_shape[-2:]
        pred = batch_fn(torch.as_tensor(np.asarray(mu)[:n]), (h, w), labels).cpu().numpy().astype(int)
        if contour_samples is None:
            return pred, None
        from cu_hip import ops
        t_e, t_a = contour_samples.shape[1], contour_samples.shape[2]
        flat = torch.as_tensor(contour_samples).reshape(n * t_e * t_a, *contour_samples.shape[3:])
        masks, packed = batch_fn(flat, (h, w), labels, packed=True)
        self._entropy_map = ops.mask_entropy(packed, n, w, mean=False)[1].cpu().numpy()
        # 0/1 like the reference's (int64 there, one byte here: a predict step holds N * T_e * T_a full-size masks)
        return pred, masks.cpu().numpy().reshape(n, t_e, t_a, *image_shape[1:])

    def _bind_datamodule_fns(self):
        dm = getattr(getattr(self, "trainer", None), "datamodule", None)
        if dm is not None:
            self.umap_fn = staticmethod(dm.umap_fn)
            self.contour_to_mask_fn = staticmethod(dm.contour_to_mask_fn)
            self.skew_umap_fn = staticmethod(dm.skew_umap_fn)

    def on_predict_start(self):
        self._bind_datamodule_fns()

    def on_fit_start(self):
        super().on_fit_start()
        self._bind_datamodule_fns()

    def predict_step(self, batch: Any, batch_idx: int, dataloader_idx: int = 0):
        """reference contour_uncertainty.py:71-130: point / instant uncertainty summaries of a BatchResult."""
        res = self._predict_step(batch)
        n = res.img.shape[0]
        if res.pred_samples is not None:
            if self._entropy_map is not None:
                res.entropy_map, self._entropy_map = self._entropy_map, None
            else:
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
