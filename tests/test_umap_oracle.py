"""CPU: projected uncertainty (oracle restatement and the package's host function) against vectors generated from the
importable reference (oracle/make_golden.py umap), and the iso-density contour construction of skew_umap."""
import warnings

import numpy as np
import pytest

from oracle import umap as U


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir / "umap_projection.npz")


def _f(a):
    return np.asarray([float(x) for x in a])


@pytest.mark.parametrize("case", range(4))
def test_projected_uncertainty_matches_reference(gold, case):
    from contour_uncertainty.utils.uncertainty_projection import projected_uncertainty
    mu, cov, alpha = gold[f"c{case}_mu"], gold[f"c{case}_cov"], gold[f"c{case}_alpha"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for lc in (0, 1):
            for fn, kw, tol in ((U.projected_uncertainty, {"every": True}, 1e-5), (projected_uncertainty, {"all": True}, 1e-6)):
                u, v, a = fn(mu, cov, alpha.copy(), linear_close=bool(lc), **kw)
                assert np.abs(_f(u) - gold[f"c{case}_lc{lc}_u"]).max() < tol * 10
                assert np.abs(v - gold[f"c{case}_lc{lc}_v"]).max() < tol
                assert np.abs(_f(a) - gold[f"c{case}_lc{lc}_a"]).max() < tol * 10
        u, v = projected_uncertainty(mu, cov, all=True)
        assert np.abs(_f(u) - gold[f"c{case}_gauss_u"]).max() < 1e-5 and np.abs(v - gold[f"c{case}_gauss_v"]).max() < 1e-6
        u, _ = projected_uncertainty(mu, cov)
        assert np.abs(_f(u) - gold[f"c{case}_ends_u"]).max() < 1e-5
        u, _ = U.projected_uncertainty(mu, cov)
        assert np.abs(_f(u) - gold[f"c{case}_ends_u"]).max() < 1e-4


def test_iso_density_contours_of_the_package_match_the_oracle(gold):
    from contour_uncertainty.utils.skew_umap import skew_umap_contours
    mu, cov, alpha = gold["c1_mu"], gold["c1_cov"], gold["c1_alpha"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m0, c0, w0 = U.skew_umap_contours(mu, cov, alpha, linear_close=True)
        m1, c1, w1 = skew_umap_contours(mu, cov, alpha, linear_close=True)
    assert c0.shape == (200, 21, 2) and np.allclose(w0, w1)
    assert np.abs(m0 - m1).max() < 1e-3 and np.abs(c0 - c1).max() < 1e-3
    # the two innermost contours hug the projected mode, the outermost are +-(2 sigma) apart at most
    assert np.abs(c0[99] - m0).max() < 1.0 and np.abs(c0[100] - m0).max() < 1.0


def test_skew_normal_mode_matches_reference(golden_dir):
    """BivariateSkewNormal.mode + delta / skewness / m0 / univariate_mode vs the imported reference
    (tests/golden/skew_mode.npz; reference distributions/bivariateskewnormal.py:73-82,195-219)."""
    import torch
    from contour_uncertainty.distributions import bivariateskewnormal as B
    g = np.load(golden_dir / "skew_mode.npz")
    a1 = torch.tensor(g["a1"])
    for name in ("delta", "skewness", "m0"):
        assert np.allclose(getattr(B, name)(a1).numpy(), g[name], rtol=2e-6, atol=1e-7), name
    assert np.allclose(B.univariate_mode(torch.tensor(3.0), torch.tensor(2.0), a1).numpy(), g["univariate_mode"], rtol=2e-6)
    for i in range(len(g["mu"])):
        m = B.BivariateSkewNormal.mode(g["mu"][i], g["cov"][i], g["alpha"][i])
        assert m.shape == (2,) and np.allclose(m.numpy(), g["mode"][i], rtol=1e-6, atol=1e-5), i


def test_marginals_keep_the_reference_signature(gold):
    """BivariateNormal / BivariateSkewNormal.marginal (closed form) against the reference-generated projections: the
    axis-0 marginal at the normal's angle is what projected_uncertainty reports."""
    import torch
    from contour_uncertainty.distributions.bivariatenormal import BivariateNormal
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal
    mu, cov, alpha = gold["c2_mu"], gold["c2_cov"], gold["c2_alpha"]
    v = gold["c2_lc0_v"]
    for k in (1, 7, 19):
        ang = torch.tensor(np.arctan2(v[k, 1], v[k, 0]))
        m, var, sk = BivariateSkewNormal.marginal(mu[k], cov[k], alpha[k], axis=0, angle=ang)
        assert abs(float(np.sqrt(var)) - gold["c2_lc0_u"][k]) < 1e-5 and abs(float(sk) - gold["c2_lc0_a"][k]) < 1e-5
        assert m == mu[k][0]
        _, var_g = BivariateNormal.marginal(mu[k], cov[k], axis=0, angle=ang)
        assert abs(float(var_g) - float(var)) < 1e-6
        # axis 1 of the frame turned by angle == axis 0 of the frame turned a quarter turn further
        _, var1, sk1 = BivariateSkewNormal.marginal(mu[k], cov[k], alpha[k], axis=1, angle=ang)
        _, var0, sk0 = BivariateSkewNormal.marginal(mu[k], cov[k], alpha[k], axis=0, angle=ang + np.pi / 2)
        assert abs(float(var1) - float(var0)) < 1e-4 and abs(float(sk1) - float(sk0)) < 1e-5
