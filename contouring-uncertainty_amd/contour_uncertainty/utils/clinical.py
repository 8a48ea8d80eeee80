"""``contour_uncertainty.utils.clinical`` -- clinical measures of contours and masks (reference
contour_uncertainty/utils/clinical.py:11-96), with the Monte-Carlo sample sets measured on the device.

Same function names, arguments and return values as the reference.  What changes is where the work runs when it is done
thousands of times per frame (1024 sampled contours per frame feed FAC / GLS / area distributions):

  * ``perimeter`` / ``global_longitudinal_strain`` / ``compute_gls``: the length of ``contour_spline(contour, n=1001)``
    (reference utils/contour.py:9-25, scipy ``splprep(k=3, s=0)`` + ``splev``) comes from ``cu_contour_measures`` -- the FITPACK
    interpolation restated in float64 in csrc/masks.hip, one workgroup per contour -- instead of a Python loop over
    ``scipy.spatial.distance.euclidean``;
  * ``lv_area`` of a mask is a pixel count (reference vital/vital/utils/image/measure.py:21-40): NumPy for masks that are
    already arrays; ``contour_measures`` counts the pixels of the mask a contour WOULD rasterise to (``cu_contour_masks``'s
    spline + closing line + fill) without writing the mask, for whole sample sets in one launch.
"""
from __future__ import annotations

from typing import Literal, Tuple

import numpy as np
import torch

LV = 1          # vital.data.camus.config.Label.LV


def lv_area(mask, voxelarea=None):
    """EchoMeasure.structure_area(mask, Label.LV, voxelarea): pixels with the LV label, per mask of a ([N], H, W) batch."""
    m = np.asarray(mask.detach().cpu() if torch.is_tensor(mask) else mask)
    return np.isin(m, LV).sum((-2, -1)) * (1 if voxelarea is None else voxelarea)


def lv_FAC(ed_mask: np.ndarray, es_mask: np.ndarray) -> float:
    """fractional area change (ED_area - ES_area) / ED_area   (reference utils/clinical.py:11-30)"""
    ed_area = lv_area(ed_mask)
    es_area = lv_area(es_mask)
    return (ed_area - es_area) / ed_area


def contour_measures(contours, shape: Tuple[int, int] = (256, 256), round_landmarks: bool = True, area: bool = True,
                     length: bool = True):
    """contours (..., K, 2) as (x, y) pixels, array or tensor -> (areas (...) int64 | None, spline lengths (...) float64 | None) as
    NumPy arrays; ONE ``cu_contour_measures`` launch for the whole set.  ``round_landmarks``: the area is that of the mask
    ``USContourToMask`` draws (landmarks rounded first, reference data/camus/utils.py:31-45); the length always uses the
    landmarks as given, like ``perimeter``."""
    from cu_hip import ops
    c = torch.as_tensor(np.asarray(contours) if not torch.is_tensor(contours) else contours, dtype=torch.float32)
    lead = tuple(c.shape[:-2])
    c = c.reshape(-1, c.shape[-2], 2)
    if not c.is_cuda:
        c = c.cuda()
    h, w = int(shape[0]), int(shape[1])
    ar = ln = None
    if area:
        ar, _ = ops.contour_measures(c, h, w, round_landmarks=round_landmarks, area=True, length=False)
        ar = ar.cpu().numpy().astype(np.int64).reshape(lead)
    if length:
        _, ln = ops.contour_measures(c, h, w, round_landmarks=False, area=False, length=True)
        ln = ln.cpu().numpy().astype(np.float64).reshape(lead)
    return ar, ln


def perimeter(contours):
    """contours ([N], K, 2) -> length of the 1001-point interpolating spline, float or (N,) array (reference :33-49)"""
    c = np.asarray(contours.detach().cpu() if torch.is_tensor(contours) else contours, dtype=np.float32)
    _, ln = contour_measures(c, area=False)
    return float(ln) if c.ndim == 2 else ln


def _polyline_length(c: np.ndarray) -> float:
    d = np.diff(np.asarray(c, dtype=np.float64), axis=0)
    return float(np.sqrt((d * d).sum(-1)).sum())


def global_longitudinal_strain(ed_contour: np.ndarray, es_contour: np.ndarray, spline: bool = True) -> float:
    """(ED length - ES length) / ED length   (reference :52-72)"""
    if spline:
        ed_len, es_len = perimeter(np.stack([np.asarray(ed_contour), np.asarray(es_contour)]))
    else:
        ed_len, es_len = _polyline_length(ed_contour), _polyline_length(es_contour)
    return (ed_len - es_len) / ed_len


def compute_gls(frames):
    """GLS (%) of every frame of a sequence of contours w.r.t. the first one (reference :75-81)"""
    lengths = perimeter(frames)
    return ((lengths - lengths[0]) / lengths[0]) * 100


def compute_FAC(frames):
    """area change (%) of every mask of a sequence w.r.t. the first one (reference :84-90)"""
    areas = lv_area(frames, voxelarea=None)
    return ((areas - areas[0]) / areas[0]) * 100


def metric_error(prediction: float, gt: float, type: Literal["absolute", "relative"] = "absolute") -> float:  # noqa: A002
    error = np.abs(prediction - gt)
    if type == "relative":
        error /= gt
    return error
