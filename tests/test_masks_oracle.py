"""CPU: the mask oracle against scipy (the third-party code the reference calls) and basic properties."""
import numpy as np
import pytest

from oracle import masks as M


def _contour(seed, k=21, size=256):
    g = np.random.default_rng(seed)
    t = np.linspace(0.0, np.pi, k)
    c = size / 2
    rx, ry = (0.16 + 0.19 * g.random()) * size, (0.16 + 0.19 * g.random()) * size
    x = c + rx * np.cos(t) + g.normal(size=k) * 2
    y = c - ry * np.sin(t) + 0.15 * size + g.normal(size=k) * 2
    return np.stack([x, y], -1).clip(1, size - 2)


@pytest.mark.parametrize("seed", range(5))
def test_fitpack_restatement_matches_scipy(seed):
    pts = _contour(seed).round()
    ref = M.contour_spline(pts, n=1000)
    got = M.fitpack_interp(pts, n=1000)
    assert np.abs(got - ref).max() < 1e-6
    assert np.allclose(got[0], pts[0]) and np.allclose(got[-1], pts[-1])


def test_reconstruction_properties():
    pts = _contour(1)
    m = M.us_contour_to_mask(pts)
    assert m.shape == (256, 256) and set(np.unique(m)) == {0, 1}
    ij = pts.round().astype(int)
    assert m[ij[:, 1], ij[:, 0]].all()                       # the landmarks are on the mask
    assert m[0].sum() == 0 and m[:, 0].sum() == 0            # nothing leaks to the border
    area = m.sum()
    assert 0.02 * 256 * 256 < area < 0.5 * 256 * 256
    # duplicate points: scipy refuses, the reference falls back to the raw points (still a closed, filled polygon edge)
    dup = pts.copy()
    dup[5] = dup[4]
    assert M.us_contour_to_mask(dup).sum() > 0


def test_sample_entropy():
    s = np.zeros((4, 1, 8, 8))
    s[:2, 0, 2:4, 2:4] = 1
    s[:, 0, 5, 5] = 1
    u = M.sample_entropy(s)
    assert np.isclose(u[2, 2], 1.0) and u[5, 5] == 0 and u[0, 0] == 0
