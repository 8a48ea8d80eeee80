"""Oracle (test infrastructure): sampled contour -> filled mask -> entropy map, restated from the reference's text.

  * ``contour_spline``   reference contour_uncertainty/utils/contour.py:9-25   (scipy splprep k=3 s=0 + splev, n points;
                         on any scipy error the raw points are used)
  * ``reconstruction``   reference .../utils/contour.py:28-40                   (spline n=1000, round, upper clip only,
                         closing ``skimage.draw.line`` last -> first point, ``binary_fill_holes``)
  * ``sample_entropy``   reference contour_uncertainty/task/uncertainty.py:107-133
  * ``us_contour_to_mask`` reference contour_uncertainty/data/camus/utils.py:31-45,94-.. (LV-only branch: landmarks rounded
                         to integers first)

utils/contour.py cannot be imported here (it needs scikit-image, which is absent); what it calls is third-party code
that IS present (scipy 1.15: ``interpolate.splprep/splev``, ``ndimage.binary_fill_holes``), so the restatement calls
the same functions with the same arguments, and ``skimage.draw.line`` is the integer Bresenham walk of ``_line``.
``fitpack_interp`` restates what ``splprep(k=3, s=0)`` computes (FITPACK parcur: chord-length parameters, interior
knots at u[2..m-3], interpolation) so that an implementation can be checked piece by piece; it is pinned to scipy in
tests/test_masks_oracle.py.
"""
from __future__ import annotations

import numpy as np
from scipy import interpolate
from scipy.ndimage import binary_fill_holes


def _line(r0: int, c0: int, r1: int, c1: int):
    """skimage.draw.line (Bresenham, both end points)."""
    dr, dc = abs(r1 - r0), abs(c1 - c0)
    sr, sc = (1 if r1 >= r0 else -1), (1 if c1 >= c0 else -1)
    steep = dr > dc
    if steep:
        r0, c0, r1, c1, dr, dc, sr, sc = c0, r0, c1, r1, dc, dr, sc, sr
    d = 2 * dr - dc
    rr, cc = [], []
    r, c = r0, c0
    for _ in range(dc + 1):
        if steep:
            rr.append(c), cc.append(r)
        else:
            rr.append(r), cc.append(c)
        while d >= 0 and dc > 0:
            r += sr
            d -= 2 * dc
        c += sc
        d += 2 * dr
    return np.array(rr), np.array(cc)


def contour_spline(mu: np.ndarray, n: int = 1001, close: bool = False) -> np.ndarray:
    try:
        tck, _ = interpolate.splprep([mu[:, 0], mu[:, 1]], k=3, s=0)
        spline = np.array(interpolate.splev(np.linspace(0, 1.0, n), tck)).transpose()
    except Exception:       # the reference has a bare `except:`
        spline = mu
    if close:
        spline = np.concatenate((spline, spline[0][None]))
    return spline


def reconstruction(points: np.ndarray, height: int, width: int) -> np.ndarray:
    seg = np.zeros((height, width))
    spline = contour_spline(points, n=1000).round().astype(int)
    seg[spline[:, 1].clip(max=height - 1), spline[:, 0].clip(max=width - 1)] = 1     # negative indices wrap (numpy)
    pts = points.round().astype(int)
    rr, cc = _line(pts[-1, 1], pts[-1, 0], pts[0, 1], pts[0, 0])
    seg[rr.clip(max=height - 1, min=0), cc.clip(max=width - 1, min=0)] = 1
    return binary_fill_holes(seg).astype(int)


def us_contour_to_mask(landmarks: np.ndarray, shape=(256, 256)) -> np.ndarray:
    """USContourToMask.__call__, LV-only labels, reconstruction_type='spline'."""
    return reconstruction(np.asarray(landmarks).round().astype(int).squeeze(), shape[0], shape[1])


def sample_entropy(samples: np.ndarray) -> np.ndarray:
    """samples (S, 1, H, W) of 0/1 -> binary entropy (base 2) of the mean map; non-finite -> 0."""
    import scipy.stats
    y = samples.mean(0)
    y = np.concatenate([y, 1 - y], axis=0)
    u = scipy.stats.entropy(y, axis=0, base=2)
    u[~np.isfinite(u)] = 0
    return u


# ---------------------------------------------------------------------------------------------------------------------
def fitpack_interp(points: np.ndarray, n: int = 1000) -> np.ndarray:
    """What splprep(k=3, s=0) + splev(linspace(0,1,n)) compute, in plain numpy (f64).  points (m, 2), m >= 4."""
    p = np.asarray(points, dtype=np.float64)
    m = len(p)
    d = np.sqrt(((p[1:] - p[:-1]) ** 2).sum(1))
    if m < 4 or (d <= 0).any():
        raise ValueError("invalid input for an interpolating cubic spline")
    u = np.concatenate([[0.0], np.cumsum(d)]) / d.sum()
    t = np.concatenate([[u[0]] * 4, u[2:m - 2], [u[-1]] * 4])        # m + 4 knots

    def basis_row(x):
        """non-zero cubic B-splines at x: (first index j, 4 values N_j..N_{j+3})"""
        l = np.searchsorted(t, x, side="right") - 1
        l = min(max(l, 3), m - 1)                                     # span t[l] <= x < t[l+1], clamped to the last one
        N = np.zeros(4)
        N[0] = 1.0
        for deg in range(1, 4):
            saved = 0.0
            for r in range(deg):
                tr, tl = t[l + r + 1], t[l + 1 - deg + r]
                term = N[r] / (tr - tl)
                N[r] = saved + (tr - x) * term
                saved = (x - tl) * term
            N[deg] = saved
        return l - 3, N

    A = np.zeros((m, m))
    for i, ui in enumerate(u):
        j, N = basis_row(ui)
        A[i, j:j + 4] = N
    c = np.linalg.solve(A, p)
    out = np.zeros((n, 2))
    for q, x in enumerate(np.linspace(0, 1.0, n)):
        j, N = basis_row(x)
        out[q] = N @ c[j:j + 4]
    return out
