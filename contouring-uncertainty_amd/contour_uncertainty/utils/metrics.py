"""Dice score (reference utils/metrics.py:9-42 wraps ``medpy.metric.dc``): 2|A & B| / (|A| + |B|) per label, averaged."""
from __future__ import annotations

import numpy as np


def dc(result: np.ndarray, reference: np.ndarray) -> float:
    result = np.atleast_1d(result.astype(bool))
    reference = np.atleast_1d(reference.astype(bool))
    inter = np.count_nonzero(result & reference)
    size = np.count_nonzero(result) + np.count_nonzero(reference)
    return 2.0 * inter / float(size) if size else 0.0


class Dice:
    def __init__(self, labels=None, exclude_bg: bool = True):
        self.labels = labels

    def __call__(self, pred: np.ndarray, target: np.ndarray) -> float:
        ids = [int(l) for l in (self.labels or [0, 1])]
        ids = [l for l in ids if l != 0] or [1]
        return float(np.mean([dc(pred == l, target == l) for l in ids]))
