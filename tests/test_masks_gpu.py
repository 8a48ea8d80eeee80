"""GPU contour -> mask -> entropy kernels (cu_contour_masks / cu_mask_entropy) against the scipy-based oracle.

Masks are integer work: the bar is bit-exact.  The only source of a difference is a spline point that lands within
~1e-9 px of a rounding boundary (the collocation solve is LU in scipy, banded elimination here); the tests allow a
mask to differ in at most 2 pixels and require almost all of them to be identical."""
import numpy as np
import pytest
import torch

from oracle import masks as M

pytestmark = pytest.mark.gpu


def _contour(seed, k=21, size=256, noise=2.0):
    g = np.random.default_rng(seed)
    t = np.linspace(0.0, np.pi, k)
    c = size / 2
    rx, ry = (0.16 + 0.19 * g.random()) * size, (0.16 + 0.19 * g.random()) * size
    x = c + rx * np.cos(t) + g.normal(size=k) * noise
    y = c - ry * np.sin(t) + 0.15 * size + g.normal(size=k) * noise
    return np.stack([x, y], -1)


def _gpu_masks(pts, h, w, rounded):
    from cu_hip import ops
    packed, by = ops.contour_masks(torch.tensor(pts, dtype=torch.float32).cuda(), h, w, round_landmarks=rounded)
    torch.cuda.synchronize()
    return packed.cpu().numpy(), by.cpu().numpy()


def _unpack(packed, w):
    bits = (packed.astype(np.uint32)[..., None] >> np.arange(32, dtype=np.uint32)) & 1
    return bits.reshape(*packed.shape[:-1], 256)[..., :w].astype(np.uint8)


def _check(pts, h, w, rounded, max_px=2, min_exact=0.9):
    packed, by = _gpu_masks(pts, h, w, rounded)
    assert (_unpack(packed, w) == by).all()
    exact = 0
    for i, p in enumerate(pts.astype(np.float32)):
        ref = M.us_contour_to_mask(p, (h, w)) if rounded else M.reconstruction(p, h, w)
        diff = int((ref != by[i]).sum())
        assert diff <= max_px, (i, diff)
        exact += diff == 0
    assert exact >= min_exact * len(pts)
    return by


@pytest.mark.parametrize("rounded", [True, False])
def test_masks_match_reference_reconstruction(rounded):
    pts = np.stack([_contour(s).clip(1, 254) for s in range(48)])
    by = _check(pts, 256, 256, rounded)
    assert 0.02 < by.mean() < 0.5


def test_masks_noisy_self_intersecting_contours():
    """MC samples can fold over themselves: the fill is a flood of the background, not a polygon rule."""
    pts = np.stack([_contour(100 + s, noise=12.0).clip(1, 254) for s in range(32)])
    _check(pts, 256, 256, False)


def test_masks_out_of_image_points_clip_and_wrap():
    """Upper clip for the spline points, numpy's negative-index wrap, min/max clip for the closing line."""
    pts = np.stack([_contour(200 + s) for s in range(16)])
    pts[:8] += np.array([70.0, 60.0])          # beyond the right / bottom edge: clipped
    pts[8:] -= np.array([75.0, 0.0])           # left of the image: indices wrap to the other side
    assert pts[:8].max() > 256 and pts[8:, :, 0].min() < 0
    _check(pts, 256, 256, False)


def test_masks_duplicate_landmarks_fall_back_to_raw_points():
    pts = np.stack([_contour(300 + s).clip(1, 254) for s in range(6)])
    pts[0, 5] = pts[0, 4]
    pts[1, 1] = pts[1, 0]
    pts[2, 20] = pts[2, 19]
    pts[3] = np.round(pts[3]); pts[3, 9] = pts[3, 8]
    _check(pts, 256, 256, True, max_px=0, min_exact=1.0)
    _check(pts, 256, 256, False, max_px=0, min_exact=1.0)


@pytest.mark.parametrize("h,w,k", [(64, 64, 21), (128, 96, 11), (200, 256, 32), (33, 70, 4), (256, 256, 3)])
def test_masks_other_sizes(h, w, k):
    g = np.random.default_rng(k)
    pts = []
    for s in range(12):
        c = _contour(400 + s, k=k, size=256, noise=1.0)
        c = c / 256.0 * np.array([w, h])
        pts.append(c.clip(0, [w - 1, h - 1]))
    _check(np.stack(pts), h, w, False)
    _check(np.stack(pts) + g.normal(size=(12, k, 2)), h, w, True)


def test_entropy_matches_sample_entropy():
    from cu_hip import ops
    f, s = 3, 25
    pts = np.stack([_contour(7 * fr, noise=0.0) + np.random.default_rng(fr * 100 + i).normal(size=(21, 2)) * 3.0
                    for fr in range(f) for i in range(s)]).clip(1, 254)
    packed, by = ops.contour_masks(torch.tensor(pts, dtype=torch.float32).cuda(), 256, 256)
    mean, ent = ops.mask_entropy(packed, f, 256)
    torch.cuda.synchronize()
    by = by.cpu().numpy().reshape(f, s, 1, 256, 256)
    for fr in range(f):
        ref = M.sample_entropy(by[fr].astype(np.float64))
        assert np.allclose(mean[fr].cpu().numpy(), by[fr, :, 0].mean(0), atol=1e-6)
        assert np.allclose(ent[fr].cpu().numpy(), ref, atol=1e-5)
        assert ref.max() > 0.9


def test_masks_reject_bad_sizes():
    from cu_hip import ops
    from cu_hip.lib import ContourHipError
    with pytest.raises(ContourHipError):
        ops.contour_masks(torch.zeros(1, 40, 2).cuda(), 256, 256)
    with pytest.raises(ContourHipError):
        ops.contour_masks(torch.zeros(1, 21, 2).cuda(), 300, 256)


def test_masks_edge_cases():
    """Single contour, two / three landmarks (no spline possible), collinear landmarks, everything outside the image,
    non-finite coordinates (must not fault: the reference would raise an IndexError there)."""
    from cu_hip import ops
    # K = 2 and K = 3: raw landmarks + closing line, like the reference's fallback
    for k in (2, 3):
        pts = np.array([[[10.2, 12.7], [40.1, 30.3], [22.0, 50.9]][:k]], dtype=np.float32)
        _check(pts, 64, 64, False, max_px=0, min_exact=1.0)
    # collinear, distinct landmarks: the spline is the segment, the closing line retraces it, nothing to fill
    t = np.linspace(0, 1, 21, dtype=np.float32)
    line = np.stack([20 + 200 * t, 30 + 150 * t], -1)[None]
    by = _check(line, 256, 256, False, max_px=0, min_exact=1.0)
    assert 150 < by.sum() < 700
    # far outside the image on the high side: every point clips onto the last row / column
    far = (_contour(1) + 400.0)[None]
    _check(far, 256, 256, False, max_px=0, min_exact=1.0)
    # NaN / inf: no fault, no pixel from the bad coordinates beyond the image
    bad = _contour(2)[None].astype(np.float32)
    bad[0, 3] = np.nan
    bad[0, 7, 0] = np.inf
    _, m = ops.contour_masks(torch.tensor(bad).cuda(), 256, 256)
    torch.cuda.synchronize()
    assert m.shape == (1, 256, 256) and int(m.max()) <= 1
    # one sample per frame: entropy is zero everywhere, the mean is the mask
    pk, m1 = ops.contour_masks(torch.tensor(_contour(3)[None], dtype=torch.float32).cuda(), 256, 256)
    mean, ent = ops.mask_entropy(pk, 1, 256)
    assert float(ent.abs().max()) == 0.0 and torch.equal(mean[0].cpu(), m1[0].float().cpu())
