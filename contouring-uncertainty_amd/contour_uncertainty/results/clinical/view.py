"""Per-view clinical metrics (ED / ES pair) over the Monte-Carlo sample sets (reference
contour_uncertainty/results/clinical/view.py:16-185): fractional area change and global longitudinal strain, each with its
Monte-Carlo distribution split into aleatoric and epistemic parts.

The reference loops ``lv_FAC`` / ``global_longitudinal_strain`` over T_e x T_a samples on the host; here the areas / spline
lengths of the whole sample set of a view come from one ``cu_contour_measures`` launch each (utils/clinical.py).
The mask-based ``EchoMeasure.gls`` columns (``GLS_pred`` / ``GLS_gt``: myocardium border tracking of `vital`, out of this
path's scope) are NaN, which is also what the reference records when that routine raises (view.py:96-104)."""
from typing import List

import numpy as np
import pandas as pd

from contour_uncertainty._compat import prefix
from contour_uncertainty.data.config import BatchResult
from contour_uncertainty.results.clinical.instant import sample_areas
from contour_uncertainty.results.clinical.utils import aleatoric_epistemic_uncertainty
from contour_uncertainty.utils.clinical import contour_measures, global_longitudinal_strain, lv_FAC, metric_error

ED, ES = "ED", "ES"


def _instant(instants, key):
    for k, v in instants.items():
        if str(getattr(k, "value", k)) == key:
            return v
    raise KeyError(key)


class ViewMetric:
    PREFIX: str = ""

    def compute(self, view: BatchResult):
        raise NotImplementedError

    def __call__(self, view_results: List[BatchResult]) -> pd.DataFrame:
        return pd.DataFrame({view.id: prefix(self.compute(view), self.PREFIX) for view in view_results}).T


class FAC(ViewMetric):
    PREFIX = "FAC_"
    MIN_VALUE = 0
    MAX_VALUE = 1

    def compute(self, view: BatchResult):
        ed, es = _instant(view.instants, ED), _instant(view.instants, ES)
        pred = lv_FAC(view.pred[ed], view.pred[es])
        gt = lv_FAC(view.gt[ed], view.gt[es])
        areas = sample_areas(view).astype(float)                   # (F, T_e, T_a)
        with np.errstate(divide="ignore", invalid="ignore"):
            mc = (areas[ed] - areas[es]) / areas[ed]               # (T_e, T_a)
        sample_reject = np.logical_or(mc < self.MIN_VALUE, mc > self.MAX_VALUE)
        mc[sample_reject] = np.nan
        metric_mean, aleatoric_var, epistemic_var, metric_variance = aleatoric_epistemic_uncertainty(mc)
        reject = not (self.MIN_VALUE < pred <= self.MAX_VALUE)
        if np.sum(sample_reject) / np.size(sample_reject) > 0.5:
            reject = True
        error = metric_error(metric_mean, gt)
        return {"pred": pred, "gt": gt, "error": error, "mc": mc.tolist(), "std": metric_variance, "mean": metric_mean,
                "aleatoric_std": aleatoric_var, "epistemic_std": epistemic_var, "reject": reject,
                "sample_reject": sample_reject}


class GLS(ViewMetric):
    PREFIX = "GLS_"
    MIN_VALUE = 0
    MAX_VALUE = 1

    def compute(self, view: BatchResult):
        ed, es = _instant(view.instants, ED), _instant(view.instants, ES)
        output = {}
        if view.contour is not None:
            contour_pred = global_longitudinal_strain(view.mu[ed], view.mu[es])
            contour_gt = global_longitudinal_strain(view.contour[ed], view.contour[es])
            # (the reference indexes `pred_samples` here (view.py:131-135), i.e. feeds MASKS to a contour routine and lands in
            #  its bare `except`; the quantity the column is named for is the strain of the sampled CONTOURS)
            _, lengths = contour_measures(view.contour_samples, area=False)          # (F, T_e, T_a)
            mc = (lengths[ed] - lengths[es]) / lengths[ed]
            metric_mean, aleatoric_var, epistemic_var, metric_variance = aleatoric_epistemic_uncertainty(mc)
            error = metric_error(metric_mean, contour_gt)
            output = {"contour_pred": contour_pred, "contour_gt": contour_gt, "contour_error": error,
                      "contour_mc": mc.tolist(), "contour_std": metric_variance, "contour_mean": metric_mean,
                      "contour_aleatoric_std": aleatoric_var, "contour_epistemic_std": epistemic_var}
        output.update({"pred": np.nan, "gt": np.nan, "reject": True})
        return output
