"""Per-instant clinical metrics over the Monte-Carlo sample sets (reference contour_uncertainty/results/clinical/instant.py)."""
from typing import List

import numpy as np
import pandas as pd

from contour_uncertainty._compat import prefix
from contour_uncertainty.data.config import BatchResult
from contour_uncertainty.results.clinical.utils import aleatoric_epistemic_uncertainty
from contour_uncertainty.utils.clinical import contour_measures, lv_area, metric_error


def sample_areas(view: BatchResult) -> np.ndarray:
    """LV area in pixels of every sampled segmentation of a view, (F, T_e, T_a): from ``pred_samples`` when the step kept the
    masks, else from ``contour_samples`` on the device (the same rasterisation, counted without writing the masks)."""
    if view.pred_samples is not None:
        return lv_area(view.pred_samples).astype(np.int64)
    shape = np.asarray(view.gt).shape[-2:] if view.gt is not None else np.asarray(view.pred).shape[-2:]
    return contour_measures(view.contour_samples, shape, length=False)[0]


class InstantMetric:
    PREFIX: str = ""

    def compute(self, view: BatchResult, instant_key: str, instant: int):
        raise NotImplementedError

    def __call__(self, view_results: List[BatchResult]) -> pd.DataFrame:
        res = {}
        for view in view_results:
            for instant_key, instant in view.instants.items():
                res[f"{view.id}/{instant_key}"] = prefix(self.compute(view, instant_key, instant), self.PREFIX)
        return pd.DataFrame(res).T


class AreaError(InstantMetric):
    """reference instant.py:29-58"""
    PREFIX = "Area_"

    def compute(self, view: BatchResult, instant_key: str, instant: int):
        voxelspacing = np.prod(view.voxelspacing)
        area_pred = lv_area(view.pred[instant]) * voxelspacing
        area_gt = lv_area(view.gt[instant]) * voxelspacing
        area_mc = sample_areas(view)[instant].astype(float) * voxelspacing           # (T_e, T_a)
        metric_mean, aleatoric_var, epistemic_var, metric_variance = aleatoric_epistemic_uncertainty(area_mc)
        error = metric_error(metric_mean, area_gt)
        return {"pred": area_pred, "gt": area_gt, "error": error, "mc": area_mc.tolist(), "std": metric_variance,
                "mean": metric_mean, "aleatoric_std": aleatoric_var, "epistemic_std": epistemic_var}
