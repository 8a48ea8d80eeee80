"""Oracle (test infrastructure): clinical measures, restated from the reference's text with the SciPy / NumPy routines the
reference itself calls.

  * ``contour_spline``                     reference contour_uncertainty/utils/contour.py:9-25 (splprep k=3 s=0, splev at n points,
                                           the raw landmarks when splprep raises)
  * ``perimeter`` / ``global_longitudinal_strain`` / ``compute_gls``   reference contour_uncertainty/utils/clinical.py:33-81
  * ``lv_area`` / ``lv_FAC`` / ``compute_FAC``                         reference utils/clinical.py:11-30,84-90 over
                                           EchoMeasure.structure_area, reference vital/vital/utils/image/measure.py:21-40
  * ``aleatoric_epistemic_uncertainty``    reference contour_uncertainty/results/clinical/utils.py:3-20

PARITY: ``aleatoric_epistemic_uncertainty`` is pinned by tests/golden/clinical.npz, written from the imported reference
function (oracle/make_golden.py clinical).  ``utils/clinical.py`` itself is NOT importable here (it imports skimage through
utils/contour.py and `vital`), and the reference holds no vectors for it: the other functions are **parity unpinned** --
text restatements that call the same scipy.interpolate routines on the same arguments."""
from __future__ import annotations

import numpy as np
from scipy import interpolate
from scipy.spatial import distance

LV = 1


def contour_spline(mu, n=1001, close=False):
    try:
        tck, u = interpolate.splprep([mu[:, 0], mu[:, 1]], k=3, s=0)
        unew = np.linspace(0, 1.0, n)
        spline = np.array(interpolate.splev(unew, tck)).transpose()
    except Exception:      # noqa: BLE001 -- the reference's bare except
        spline = mu
    if close:
        spline = np.concatenate((spline, spline[0][None]))
    return spline


def perimeter(contours):
    def one(c):
        c = contour_spline(c)
        return np.sum([distance.euclidean(c[i], c[i + 1]) for i in range(c.shape[0] - 1)])
    return one(contours) if contours.ndim == 2 else np.array([one(c) for c in contours])


def global_longitudinal_strain(ed_contour, es_contour, spline=True):
    if spline:
        ed_contour, es_contour = contour_spline(ed_contour), contour_spline(es_contour)
    ed_len = np.sum([distance.euclidean(ed_contour[i], ed_contour[i + 1]) for i in range(len(ed_contour) - 1)])
    es_len = np.sum([distance.euclidean(es_contour[i], es_contour[i + 1]) for i in range(len(es_contour) - 1)])
    return (ed_len - es_len) / ed_len


def compute_gls(frames):
    lengths = perimeter(frames)
    return ((lengths - lengths[0]) / lengths[0]) * 100


def lv_area(mask, voxelarea=None):
    return np.isin(mask, LV).sum((-2, -1)) * (1 if voxelarea is None else voxelarea)


def lv_FAC(ed_mask, es_mask):
    ed_area, es_area = lv_area(ed_mask), lv_area(es_mask)
    return (ed_area - es_area) / ed_area


def compute_FAC(frames):
    areas = lv_area(frames)
    return ((areas - areas[0]) / areas[0]) * 100


def aleatoric_epistemic_uncertainty(metric_mc):
    assert metric_mc.ndim == 2
    metric_means = np.nanmean(metric_mc, axis=-1)
    metric_vars = np.nanstd(metric_mc, axis=-1)
    metric_mean = np.nanmean(metric_means)
    epistemic_var = np.nanstd(metric_means)
    aleatoric_var = np.nanmean(metric_vars)
    return metric_mean, aleatoric_var, epistemic_var, epistemic_var + aleatoric_var
