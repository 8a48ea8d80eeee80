"""Bivariate normal / skew-normal helpers (a9, a10 of SURVEY 8a) vs golden vectors from the imported reference."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_logpdf_grids_vs_reference_golden(golden_dir):
    from contour_uncertainty.distributions.bivariatenormal import BivariateNormal
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal
    g = np.load(golden_dir / "nll_heads.npz")
    pos = torch.from_numpy(g["grid_pos"]).cuda()
    loc, cov, al = (torch.from_numpy(g[k]).cuda() for k in ("grid_loc", "grid_cov", "grid_alpha"))
    lp = BivariateNormal.logpdf(pos, loc, cov).cpu()
    assert lp.shape == (64, 64)
    assert torch.allclose(lp, torch.from_numpy(g["grid_gauss_logpdf"]), rtol=1e-4, atol=2e-4)
    ls = BivariateSkewNormal.logpdf(pos, loc, cov, al).cpu()
    ref = torch.from_numpy(g["grid_skew_logpdf"])
    # log(Phi + 1e-7) cancels in f32 where Phi is tiny (the reference's own value is noisy there): compare where Phi > 1e-3
    ok = (ref - torch.from_numpy(g["grid_gauss_logpdf"]) - math.log(2)) > math.log(1e-3)
    assert torch.allclose(ls[ok], ref[ok], rtol=1e-4, atol=5e-4)
    assert torch.allclose(ls[~ok], ref[~ok], rtol=0, atol=0.5)
    # the reference's manual KAT (bivariatenormal.py:98-103): pdf vs scipy.stats.multivariate_normal
    pts = torch.from_numpy(g["kat_pts"]).cuda()
    pdf = BivariateNormal.pdf(pts, torch.tensor([100.0, 100.0]).cuda(), torch.tensor([[25.0, 4.0], [4.0, 50.0]]).cuda())
    assert np.allclose(pdf.cpu().numpy(), g["kat_scipy"], rtol=1e-4)


def test_per_point_nll_terms_vs_reference_golden(golden_dir):
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal
    g = np.load(golden_dir / "nll_heads.npz")
    mu, y, cov, alpha = (torch.from_numpy(g[k]).cuda() for k in ("mu", "y", "cov", "alpha"))
    nll, t1, t2, t3 = BivariateSkewNormal.nll(y, mu, cov, alpha)
    cdf = torch.from_numpy(np.exp(g["skew_t3"]) - 1e-7)
    ok = cdf > 1e-2
    for got, key in ((nll, "skew_nll"), (t1, "skew_t1"), (t2, "skew_t2"), (t3, "skew_t3")):
        ref = torch.from_numpy(g[key])
        assert torch.allclose(got.cpu()[ok], ref[ok], rtol=1e-4, atol=1e-4), key


def test_skew_rvs_moments():
    """rvs_fast: mean = mu + sqrt(2/pi) delta, delta = Sigma alpha / sqrt(1 + alpha^T Sigma alpha)."""
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal
    mu = torch.tensor([100.0, 150.0]).cuda()
    cov = torch.tensor([[10.0, -5.0], [-5.0, 10.0]]).cuda()
    alpha = torch.tensor([5.0, 0.0]).cuda()
    x = BivariateSkewNormal.rvs_fast(mu, cov, alpha, size=200000, seed=7).cpu()
    assert x.shape == (200000, 2)
    delta = (cov.cpu() @ alpha.cpu()) / torch.sqrt(1 + alpha.cpu() @ cov.cpu() @ alpha.cpu())
    mean = mu.cpu() + math.sqrt(2 / math.pi) * delta
    assert torch.allclose(x.mean(0), mean, atol=0.03)
    var = cov.cpu() - (2 / math.pi) * torch.outer(delta, delta)
    assert torch.allclose(torch.cov(x.T), var, atol=0.08)
    # shared normal draws reproduce the 3-D Gaussian construction exactly
    eps = torch.randn(1, 5, 3, generator=torch.Generator().manual_seed(1))
    got = BivariateSkewNormal.rvs_fast(mu, cov, alpha, size=5, eps=eps.cuda()).cpu()
    cs = torch.zeros(3, 3)
    cs[0, 0] = 1; cs[1:, 0] = delta; cs[0, 1:] = delta; cs[1:, 1:] = cov.cpu()
    z = eps[0] @ torch.linalg.cholesky(cs).T
    ref = torch.where(z[:, :1] <= 0, -z[:, 1:], z[:, 1:]) + mu.cpu()
    assert torch.allclose(got, ref, atol=1e-3)


def test_skew_rvs_vs_reference_golden(golden_dir):
    """cu_skew_rvs against BivariateSkewNormal.rvs_fast of the imported reference with the generator state pinned
    (tests/golden/skew_grid.npz: rvs_eps are the normals MultivariateNormal.sample drew, rvs_x what it returned;
    reference distributions/bivariateskewnormal.py:159-191)."""
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal
    g = np.load(golden_dir / "skew_grid.npz")
    mu, cov, alpha = (torch.from_numpy(g[k]).cuda() for k in ("rvs_mu", "rvs_cov", "rvs_alpha"))
    eps = torch.from_numpy(g["rvs_eps"]).float()[None].cuda()           # (1, 64, 3)
    got = BivariateSkewNormal.rvs_fast(mu, cov, alpha, size=(64,), eps=eps).cpu()
    ref = torch.from_numpy(g["rvs_x"])
    assert got.shape == ref.shape == (64, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
