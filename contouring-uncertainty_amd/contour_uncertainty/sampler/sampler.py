"""``contour_uncertainty.sampler.sampler.Sampler`` (reference sampler/sampler.py:6-78): interface + point ordering."""
from __future__ import annotations

import math

import numpy as np


class Sampler:
    def __call__(self, mu, cov, n: int):
        """mu (K, 2), cov (K, 2, 2) -> sampled contours (n, K, 2)."""
        raise NotImplementedError

    @staticmethod
    def get_points_order(nb_points: int = 21, nb_initial_points: int = 3, levels: int = None):
        """Point order by repeatedly splitting in half (reference sampler.py:43-78 / psm.py:43-71).
        get_points_order(21, levels=3) = ([0, 10, 20], [[5, 15], [2, 7, 13, 18], [1, 3, 6, 8, 12, 14, 17, 19]])."""
        initial_points = np.round(np.linspace(0, nb_points - 1, nb_initial_points)).astype(int).tolist()
        levels = levels or int(math.log(nb_points, 2))
        all_points, point_order = list(initial_points), []
        for _ in range(levels):
            level_points = []
            for j in range(len(all_points) - 1):
                if all_points[j] + 1 != all_points[j + 1]:
                    point = (all_points[j] + all_points[j + 1]) / 2
                    point = math.ceil(point) if point > nb_points / 2 else math.floor(point)  # round towards the base
                    level_points.append(int(point))
            if not level_points:
                break
            all_points.extend(level_points)
            all_points.sort()
            point_order.append(level_points)
        return initial_points, point_order
