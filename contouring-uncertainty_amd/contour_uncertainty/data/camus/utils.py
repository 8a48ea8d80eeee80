"""``USContourToMask`` (reference contour_uncertainty/data/camus/utils.py:31-100), the CAMUS datamodule's
``contour_to_mask_fn`` (reference data/camus/datamodule.py:72), for the single-structure (LV endocardium) labels the
dsnt tasks of the hot path are configured with.  ``__call__`` keeps the reference signature for one contour; ``batch``
is what ``ContourUncertaintyTask.convert_to_mask`` uses: every contour of a predict step in one kernel launch.

The LV + MYO branch (reference utils.py:48-82; two structures = 2 x (2 points_per_side - 1) landmarks) is host code, as in
the reference: two SciPy spline fits, the polygon between the endocardial and the reversed epicardial spline filled by the
restated ``skimage.draw.polygon`` rule (``utils/contour.py::polygon_mask``), the LV mask from the device rasteriser.  The
u-map functors combine the two structures' maps the way the reference does (utils.py:106-147)."""
from __future__ import annotations

import numpy as np
import torch

from contour_uncertainty.utils.contour import (contour_spline, linear_reconstruction, polygon_mask, reconstruction,
                                               reconstruction_batch)
from contour_uncertainty.utils.skew_umap import skew_umap
from contour_uncertainty.utils.umap import uncertainty_map

LABEL_BG, LABEL_LV, LABEL_MYO = 0, 1, 2           # vital.data.camus.config.Label


def split_landmarks(landmarks):
    """first half = LV endocardium, second half = epicardium (reference utils.py:14-21)"""
    p1 = len(landmarks) // 2
    p2 = p1 + len(landmarks) // 2
    return landmarks[:p1], landmarks[p1:p2]


def _has_myo(labels) -> bool:
    return labels is not None and any(int(getattr(lb, "value", lb)) == LABEL_MYO for lb in labels)


class USContourToMask:
    @staticmethod
    def __call__(landmarks, shape=(256, 256), labels=None, apply_argmax: bool = True, reconstruction_type: str = "spline"):
        if reconstruction_type not in ("spline", "linear"):
            raise ValueError(reconstruction_type)
        landmarks = np.asarray(landmarks).round().astype(int).squeeze()
        assert landmarks.ndim == 2 and landmarks.shape[1] == 2
        if _has_myo(labels):           # reference utils.py:48-82
            rec = (lambda pts, h, w: linear_reconstruction(pts, (h, w)).astype(int)) if reconstruction_type == "linear" \
                else reconstruction
            lv, myo = split_landmarks(landmarks)
            lv_spline = contour_spline(lv, n=1000).round().astype(int)
            myo_spline = contour_spline(myo, n=1000).round().astype(int)
            polygon = np.concatenate([lv_spline, np.flip(myo_spline, axis=0)])
            lv_mask = rec(lv, shape[0], shape[1])
            myo_mask = polygon_mask(polygon[:, 1], polygon[:, 0], shape)
            seg_map = np.zeros((3,) + tuple(shape), dtype=int)
            seg_map[LABEL_LV] = lv_mask
            seg_map[LABEL_MYO] = np.clip(myo_mask - lv_mask, a_min=0, a_max=1)
            seg_map[LABEL_BG] = seg_map.sum(0) == 0
            return seg_map.argmax(0) if apply_argmax else seg_map
        if reconstruction_type == "linear":
            seg = linear_reconstruction(landmarks, shape).astype(int)
        else:
            seg = reconstruction_batch(landmarks[None], shape[0], shape[1])[0].cpu().numpy().astype(int)
        return seg if apply_argmax else seg[None]

    @staticmethod
    def batch(landmarks, shape=(256, 256), labels=None, packed: bool = False):
        """landmarks (M, K, 2) tensor/array -> uint8 cuda tensor (M, H, W) [, packed (M, H, 8) int32]; the landmarks are
        rounded first, like ``__call__``."""
        if _has_myo(labels):           # host branch, item by item (not on the sampled-contour fast path)
            arr = np.asarray(torch.as_tensor(landmarks).detach().cpu())
            segs = np.stack([USContourToMask.__call__(a, shape, labels) for a in arr])
            t = torch.as_tensor(segs, dtype=torch.uint8, device="cuda")
            return (t, None) if packed else t
        return reconstruction_batch(landmarks, shape[0], shape[1], round_landmarks=True, packed=packed)


class USSkewUmap:
    """reference data/camus/utils.py:126-151, single-structure labels: projected mode + skew-normal uncertainty map."""

    @staticmethod
    def __call__(mu, cov, alpha, labels=None):
        if _has_myo(labels):           # reference utils.py:129-141
            (mu_lv, mu_myo), (cov_lv, cov_myo), (a_lv, a_myo) = split_landmarks(mu), split_landmarks(cov), split_landmarks(alpha)
            lv_mode, lv_umap = skew_umap(mu_lv, cov_lv, a_lv)
            myo_mode, myo_umap = skew_umap(mu_myo, cov_myo, a_myo)
            umap = np.clip(lv_umap / lv_umap.max() + myo_umap / myo_umap.max(), a_min=0, a_max=1) / 2
            return np.concatenate([lv_mode, myo_mode], axis=0), umap
        projected_mode, umap = skew_umap(mu, cov, alpha, linear_close=True)
        return projected_mode, umap / umap.max()


class USUMap:
    """reference data/camus/utils.py:103-123, single-structure labels: Gaussian uncertainty map scaled to a maximum of 1."""

    @staticmethod
    def __call__(mu, cov, labels=None):
        if _has_myo(labels):           # reference utils.py:106-114: the skew-normal map with zero skew, per structure
            (mu_lv, mu_myo), (cov_lv, cov_myo) = split_landmarks(mu), split_landmarks(cov)
            _, lv_umap = skew_umap(mu_lv, cov_lv, np.zeros_like(mu_lv))
            _, myo_umap = skew_umap(mu_myo, cov_myo, np.zeros_like(mu_myo))
            return np.clip(lv_umap / lv_umap.max() + myo_umap / myo_umap.max(), a_min=0, a_max=1) / 2
        umap = uncertainty_map(mu, cov)
        return umap / umap.max()
