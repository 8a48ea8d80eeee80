"""GPU: skew-normal uncertainty map (200 reconstructions by cu_contour_masks + cu_mask_weighted_entropy) vs the oracle."""
import warnings

import numpy as np
import pytest
import torch

from oracle import umap as U

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [0, 2])
def test_skew_umap_matches_oracle(golden_dir, case):
    from contour_uncertainty.data.camus.utils import USSkewUmap
    from contour_uncertainty.utils.skew_umap import skew_umap
    g = np.load(golden_dir / "umap_projection.npz")
    mu, cov, alpha = g[f"c{case}_mu"], g[f"c{case}_cov"], g[f"c{case}_alpha"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mode_ref, umap_ref = U.skew_umap(mu, cov, alpha, linear_close=True)
        mode, umap = skew_umap(mu, cov, alpha, linear_close=True)
        mode2, umap_n = USSkewUmap()(mu, cov, alpha, labels=[0, 1])
    assert umap.shape == (256, 256) and np.abs(mode - mode_ref).max() < 1e-3
    # a differing mask pixel (spline point within rounding noise of .5) would show as an isolated large difference
    diff = np.abs(umap - umap_ref)
    assert (diff > 1e-4).sum() <= 4 and np.median(diff) < 1e-6
    assert 0.6 < umap_ref.max() <= np.log(2) + 1e-9
    assert np.allclose(umap_n, umap / umap.max()) and np.allclose(mode2, mode)


def test_weighted_entropy_kernel():
    from cu_hip import ops
    g = torch.Generator().manual_seed(0)
    s, h, w = 37, 40, 70
    masks = (torch.rand(s, h, w, generator=g) < 0.4)
    packed = torch.zeros(s, h, 8, dtype=torch.int64)
    for x in range(w):
        packed[:, :, x >> 5] |= masks[:, :, x].long() << (x & 31)
    packed = packed.to(torch.int32 if False else torch.int64)
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32).cuda()
    wt = torch.rand(s, generator=g)
    wt = (wt / wt.sum()).float()
    mean, ent = ops.mask_weighted_entropy(packed, 1, w, wt.cuda())
    ref = (masks.float() * wt[:, None, None]).sum(0)
    assert torch.allclose(mean[0].cpu(), ref, atol=1e-6)
    r = ref.double().clamp(1e-300, 1)
    e = -(torch.where(ref > 0, r * r.log(), torch.zeros_like(r)) + torch.where(ref < 1, (1 - r) * (1 - r).clamp_min(1e-300).log(), torch.zeros_like(r)))
    assert torch.allclose(ent[0].cpu().double(), e, atol=1e-5)


@pytest.mark.parametrize("case,close", [(0, True), (1, True), (3, False)])
def test_gaussian_uncertainty_map_matches_oracle(golden_dir, case, close):
    from contour_uncertainty.data.camus.utils import USUMap
    from contour_uncertainty.utils.umap import uncertainty_map
    g = np.load(golden_dir / "umap_projection.npz")
    mu, cov = g[f"c{case}_mu"], g[f"c{case}_cov"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = U.uncertainty_map(mu, cov, close=close)
        got = uncertainty_map(mu, cov, close=close)
        if close:
            assert np.allclose(USUMap()(mu, cov, labels=[0, 1]), got / got.max())
    assert got.shape == (256, 256) and (ref > 0).sum() > 2000
    # identical up to a handful of pixels whose spline point sits within float32 rounding of a .5 / integer boundary
    assert (np.abs(got - ref) > 1e-6).sum() <= 12
    assert abs(got.max() - ref.max()) < 1e-6
