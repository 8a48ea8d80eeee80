"""``USContourToMask`` (reference contour_uncertainty/data/camus/utils.py:31-100), the CAMUS datamodule's
``contour_to_mask_fn`` (reference data/camus/datamodule.py:72), for the single-structure (LV endocardium) labels the
dsnt tasks of the hot path are configured with.  ``__call__`` keeps the reference signature for one contour; ``batch``
is what ``ContourUncertaintyTask.convert_to_mask`` uses: every contour of a predict step in one kernel launch.

The LV + MYO branch (reference utils.py:48-82) needs ``skimage.draw.polygon`` and is outside SURVEY.md 8's scope: it
raises instead of silently producing something else."""
from __future__ import annotations

import numpy as np

from contour_uncertainty.utils.contour import linear_reconstruction, reconstruction_batch
from contour_uncertainty.utils.skew_umap import skew_umap
from contour_uncertainty.utils.umap import uncertainty_map

LABEL_MYO = 2           # vital.data.camus.config.Label.MYO


def _has_myo(labels) -> bool:
    return labels is not None and any(int(getattr(lb, "value", lb)) == LABEL_MYO for lb in labels)


class USContourToMask:
    @staticmethod
    def __call__(landmarks, shape=(256, 256), labels=None, apply_argmax: bool = True, reconstruction_type: str = "spline"):
        if reconstruction_type not in ("spline", "linear"):
            raise ValueError(reconstruction_type)
        if _has_myo(labels):
            raise NotImplementedError("USContourToMask: the LV + MYO branch is not part of this build (SURVEY.md 8)")
        landmarks = np.asarray(landmarks).round().astype(int).squeeze()
        assert landmarks.ndim == 2 and landmarks.shape[1] == 2
        if reconstruction_type == "linear":
            seg = linear_reconstruction(landmarks, shape).astype(int)
        else:
            seg = reconstruction_batch(landmarks[None], shape[0], shape[1])[0].cpu().numpy().astype(int)
        return seg if apply_argmax else seg[None]

    @staticmethod
    def batch(landmarks, shape=(256, 256), labels=None, packed: bool = False):
        """landmarks (M, K, 2) tensor/array -> uint8 cuda tensor (M, H, W) [, packed (M, H, 8) int32]; the landmarks are
        rounded first, like ``__call__``."""
        if _has_myo(labels):
            raise NotImplementedError("USContourToMask: the LV + MYO branch is not part of this build (SURVEY.md 8)")
        return reconstruction_batch(landmarks, shape[0], shape[1], round_landmarks=True, packed=packed)


class USSkewUmap:
    """reference data/camus/utils.py:126-151, single-structure labels: projected mode + skew-normal uncertainty map."""

    @staticmethod
    def __call__(mu, cov, alpha, labels=None):
        if _has_myo(labels):
            raise NotImplementedError("USSkewUmap: the LV + MYO branch is not part of this build (SURVEY.md 8)")
        projected_mode, umap = skew_umap(mu, cov, alpha, linear_close=True)
        return projected_mode, umap / umap.max()


class USUMap:
    """reference data/camus/utils.py:103-123, single-structure labels: Gaussian uncertainty map scaled to a maximum of 1."""

    @staticmethod
    def __call__(mu, cov, labels=None):
        if _has_myo(labels):
            raise NotImplementedError("USUMap: the LV + MYO branch is not part of this build (SURVEY.md 8)")
        umap = uncertainty_map(mu, cov)
        return umap / umap.max()
