"""SURVEY.md 8f rank 4 (clinical half): FAC / GLS / area over the Monte-Carlo sample sets.

CPU: the oracle's ``aleatoric_epistemic_uncertainty`` and the product's against tests/golden/clinical.npz (written from the
imported reference function); the oracle's spline length against an independent dense-sampling check.  GPU: ``cu_contour_measures``
(mask area + 1001-point spline length of thousands of contours in one launch) against oracle/clinical.py (SciPy, the routines the
reference calls -- parity otherwise unpinned, see its header) and against the masks ``cu_contour_masks`` writes; the FAC / GLS /
AreaError metric classes on a synthetic view against the reference's loops restated with the oracle."""
import numpy as np
import pytest
import torch

from oracle import clinical as OC


def test_aleatoric_epistemic_split_vs_reference_golden(golden_dir):
    from contour_uncertainty.results.clinical.utils import aleatoric_epistemic_uncertainty
    g = np.load(golden_dir / "clinical.npz")
    for i in range(3):
        mc, ref = g[f"mc{i}"], g[f"res{i}"]
        assert np.allclose(np.array(OC.aleatoric_epistemic_uncertainty(mc)), ref, rtol=1e-12, atol=0)
        assert np.allclose(np.array(aleatoric_epistemic_uncertainty(mc)), ref, rtol=1e-12, atol=0)


def test_oracle_spline_length_properties():
    t = np.linspace(0, np.pi, 21)
    arc = np.stack([128 + 60 * np.cos(t), 128 - 60 * np.sin(t)], -1)            # half circle, radius 60
    assert abs(OC.perimeter(arc) - np.pi * 60) < 0.05
    assert np.allclose(OC.perimeter(np.stack([arc, arc * 0.5])), [np.pi * 60, np.pi * 30], atol=0.05)
    assert abs(OC.global_longitudinal_strain(arc, arc * 0.8) - 0.2) < 1e-3
    assert abs(OC.global_longitudinal_strain(arc, arc * 0.8, spline=False) - 0.2) < 1e-9
    gls = OC.compute_gls(np.stack([arc, arc * 0.9, arc * 0.8]))
    assert np.allclose(gls, [0, -10, -20], atol=1e-2)
    m = np.zeros((3, 16, 16), dtype=int)
    m[0, 2:10, 2:10] = 1; m[1, 2:8, 2:8] = 1; m[2, 2:6, 2:10] = 2
    assert list(OC.lv_area(m)) == [64, 36, 0] and abs(OC.lv_FAC(m[0], m[1]) - 28 / 64) < 1e-12
    assert np.allclose(OC.compute_FAC(m[:2]), [0, -43.75])


def _contours(n, k=21, size=256, seed=0):
    from oracle.step import synthetic_batch
    _, c = synthetic_batch(n, size, k, seed=seed)
    return c.numpy()


@pytest.mark.gpu
def test_contour_measures_vs_oracle_and_masks():
    from cu_hip import ops
    from oracle import masks as MO
    c = _contours(48, seed=3)
    c[5, 7] = c[5, 6]                       # duplicate consecutive landmarks: splprep raises -> raw landmark polyline
    dev = torch.from_numpy(c).cuda()
    area, length = ops.contour_measures(dev, 256, 256, round_landmarks=True)
    _, length_raw = ops.contour_measures(dev, 256, 256, round_landmarks=False, area=False)
    pk, masks = ops.contour_masks(dev, 256, 256, round_landmarks=True)
    torch.cuda.synchronize()
    assert torch.equal(area.cpu(), masks.sum((1, 2)).int().cpu())                 # exactly the mask the same kernel writes
    for i in range(0, 48, 6):
        ref_mask = MO.us_contour_to_mask(c[i], (256, 256))
        assert abs(int(area[i]) - int(ref_mask.sum())) <= 2                        # the oracle's rasterisation (scipy + fill)
    ref_len = OC.perimeter(c.astype(np.float64))
    assert np.allclose(length_raw.cpu().numpy(), ref_len, rtol=2e-6, atol=1e-4), np.abs(length_raw.cpu().numpy() - ref_len).max()
    d = np.diff(c[5].astype(np.float64), axis=0)
    assert abs(float(length_raw[5]) - np.sqrt((d * d).sum(-1)).sum()) < 1e-3      # the fallback row
    # rounded landmarks: same routine on np.round'ed input
    ref_r = OC.perimeter(np.round(c).astype(np.float64))
    assert np.allclose(length.cpu().numpy(), ref_r, rtol=2e-6, atol=1e-4)


@pytest.mark.gpu
def test_clinical_functions_keep_the_reference_signatures():
    from contour_uncertainty.utils import clinical as UC
    c = _contours(4, seed=5).astype(np.float64)
    assert abs(UC.perimeter(c[0]) - OC.perimeter(c[0])) < 1e-3
    assert np.allclose(UC.perimeter(c), OC.perimeter(c), rtol=2e-6, atol=1e-3)
    for spline in (True, False):
        assert abs(UC.global_longitudinal_strain(c[0], c[1], spline) - OC.global_longitudinal_strain(c[0], c[1], spline)) < 1e-6
    assert np.allclose(UC.compute_gls(c), OC.compute_gls(c), atol=1e-4)
    m = np.zeros((3, 16, 16), dtype=int)
    m[0, 2:10, 2:10] = 1; m[1, 2:8, 2:8] = 1
    assert list(UC.lv_area(m)) == list(OC.lv_area(m)) and UC.lv_FAC(m[0], m[1]) == OC.lv_FAC(m[0], m[1])
    assert np.allclose(UC.compute_FAC(m[:2]), OC.compute_FAC(m[:2]))
    assert UC.metric_error(0.5, 0.4, "relative") == pytest.approx(0.25) and UC.metric_error(0.5, 0.4) == pytest.approx(0.1)


@pytest.mark.gpu
def test_view_and_instant_metrics_over_a_sample_set():
    """FAC / GLS / AreaError on a synthetic ED / ES view with T_e x T_a sampled contours: the Monte-Carlo columns equal the
    reference's host loops restated with the oracle (rasterise every sample, count; spline every sample, measure)."""
    from contour_uncertainty.data.config import BatchResult
    from contour_uncertainty.results.clinical.instant import AreaError
    from contour_uncertainty.results.clinical.view import FAC, GLS
    from contour_uncertainty.utils.contour import reconstruction_batch
    from oracle import masks as MO
    te, ta, k, size = 2, 6, 21, 128
    rng = np.random.default_rng(0)
    base = _contours(2, k, size, seed=9)                              # ED, ES
    base[1] = (base[1] - size / 2) * 0.8 + size / 2                   # ES: smaller cavity
    cs = base[:, None, None] + rng.normal(0, 1.2, size=(2, te, ta, k, 2)).astype(np.float32)
    flat = torch.from_numpy(cs.reshape(-1, k, 2)).cuda()
    pred_samples = reconstruction_batch(flat, size, size, round_landmarks=True).cpu().numpy().reshape(2, te, ta, size, size)
    pred = reconstruction_batch(torch.from_numpy(base).cuda(), size, size, round_landmarks=True).cpu().numpy()
    gt = reconstruction_batch(torch.from_numpy(base + 1.0).cuda(), size, size, round_landmarks=True).cpu().numpy()
    view = BatchResult(id="patient0001/2CH", labels=[0, 1], contour=base + 1.0, gt=gt, mu=base, contour_samples=cs,
                       pred_samples=pred_samples, pred=pred, instants={"ED": 0, "ES": 1}, voxelspacing=(0.5, 0.5))
    fac = FAC().compute(view)
    ref_mc = np.array([[OC.lv_FAC(MO.us_contour_to_mask(cs[0, j, i], (size, size)), MO.us_contour_to_mask(cs[1, j, i], (size, size)))
                        for i in range(ta)] for j in range(te)])
    assert np.allclose(np.array(fac["mc"]), ref_mc, atol=2e-3)                   # <= 2 pixels per mask of a ~2000-pixel cavity
    assert abs(fac["mean"] - OC.aleatoric_epistemic_uncertainty(ref_mc)[0]) < 2e-3 and fac["reject"] is False
    assert abs(fac["pred"] - OC.lv_FAC(pred[0], pred[1])) < 1e-12
    # without the masks in the result the areas come from cu_contour_measures: same numbers
    view2 = BatchResult(**{**vars(view), "pred_samples": None})
    assert np.array_equal(np.array(FAC().compute(view2)["mc"]), np.array(fac["mc"]))
    gls = GLS().compute(view)
    ref_g = np.array([[OC.global_longitudinal_strain(cs[0, j, i].astype(np.float64), cs[1, j, i].astype(np.float64))
                       for i in range(ta)] for j in range(te)])
    assert np.allclose(np.array(gls["contour_mc"]), ref_g, atol=2e-6)
    assert abs(gls["contour_pred"] - OC.global_longitudinal_strain(base[0].astype(np.float64), base[1].astype(np.float64))) < 1e-6
    m, al, ep, tot = OC.aleatoric_epistemic_uncertainty(ref_g)
    assert abs(gls["contour_mean"] - m) < 1e-6 and abs(gls["contour_std"] - tot) < 1e-6
    area = AreaError().compute(view, "ED", 0)
    assert area["pred"] == OC.lv_area(pred[0]) * 0.25 and np.array(area["mc"]).shape == (te, ta)
    assert abs(area["mean"] - np.nanmean(OC.lv_area(pred_samples[0]) * 0.25)) < 1e-9
    df = FAC()([view])
    assert list(df.index) == ["patient0001/2CH"] and "FAC_mean" in df.columns and "FAC_sample_reject" in df.columns
