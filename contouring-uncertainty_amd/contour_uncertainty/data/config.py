"""``contour_uncertainty.data.config``: the batch / result containers at the boundary (reference data/config.py:37-102)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional, Sequence

import numpy as np

from contour_uncertainty._compat import ContourTags, Tags  # noqa: F401  (re-exported under the reference's path)


@dataclass
class BatchResult:
    id: Any = None
    labels: Optional[Sequence[Any]] = None
    img: Any = None
    contour: Optional[np.ndarray] = None
    gt: Optional[np.ndarray] = None
    mu: Optional[np.ndarray] = None
    mode: Optional[np.ndarray] = None
    cov: Optional[np.ndarray] = None
    alpha: Optional[np.ndarray] = None
    contour_samples: Optional[np.ndarray] = None
    pred_samples: Optional[np.ndarray] = None
    pred: Optional[np.ndarray] = None
    uncertainty_map: Optional[np.ndarray] = None
    entropy_map: Optional[np.ndarray] = None
    instants: Any = None
    voxelspacing: Any = None
    post_mu: Optional[np.ndarray] = None
    post_cov: Optional[np.ndarray] = None
    point_uncertainty: Optional[dict] = None
    instant_uncertainty: Optional[dict] = None
