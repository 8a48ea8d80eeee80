"""the al / ep covariance split and the per-point sample covariance of a predict step -- the product's
vectorised helpers (contour_uncertainty.utils.posterior_stats) against the loop-for-loop restatement of the reference
(oracle/predict_stats.py; aleatoric.py:88-108, aleatoric_skew.py:65-82)."""
import numpy as np
import pytest
import torch

from oracle import predict_stats as O


def _inputs(n, te, ta, k, seed):
    g = torch.Generator().manual_seed(seed)
    mu = torch.rand(n, te, k, 2, generator=g) * 200 + 20
    a = torch.randn(n, te, k, 2, 2, generator=g)
    cov = a @ a.transpose(-1, -2) * 9 + 0.5 * torch.eye(2)
    alpha = torch.randn(n, te, k, 2, generator=g) * 2
    samples = (mu[:, :, None] + torch.randn(n, te, ta, k, 2, generator=g) * 3).numpy().astype(np.float32)
    return mu, cov, alpha, samples


@pytest.mark.parametrize("shape", [(2, 1, 25, 21), (3, 4, 7, 21), (1, 2, 2, 5)])
def test_gaussian_task_statistics_match_the_reference_loops(shape):
    from contour_uncertainty.utils.posterior_stats import sample_moments_per_member, total_moments
    mu, cov, _, samples = _inputs(*shape, seed=3)
    ref = O.aleatoric_stats(mu, cov, samples)
    m, al, ep = total_moments(mu, cov)
    assert np.allclose(m, ref["mu"], rtol=1e-6) and np.allclose(al, ref["cov_al"], rtol=1e-6)
    assert np.allclose(ep, ref["cov_ep"], rtol=1e-5, atol=1e-5) and np.allclose(al + ep, ref["cov"], rtol=1e-5, atol=1e-5)
    pm, pc = sample_moments_per_member(samples)
    assert np.allclose(pm, ref["post_mu"], rtol=1e-6) and np.allclose(pc, ref["post_cov"], rtol=1e-5, atol=1e-6)
    if shape[1] == 1:        # one member: no epistemic part, the total is the member's covariance
        assert np.abs(ep).max() == 0 and np.allclose(al, cov[:, 0].numpy())


@pytest.mark.parametrize("shape", [(2, 1, 25, 21), (3, 4, 7, 21)])
def test_skew_task_statistics_match_the_reference_loops(shape):
    from contour_uncertainty.utils.posterior_stats import sample_moments_pooled, total_moments
    mu, cov, alpha, samples = _inputs(*shape, seed=4)
    ref = O.aleatoric_skew_stats(mu, cov, alpha, samples)
    m, al, ep = total_moments(mu, cov)
    assert np.allclose(m, ref["mu"], rtol=1e-6) and np.allclose(al + ep, ref["cov"], rtol=1e-5, atol=1e-5)
    pm, pc = sample_moments_pooled(samples)
    assert np.allclose(pm, ref["post_mu"], rtol=1e-6) and np.allclose(pc, ref["post_cov"], rtol=1e-5, atol=1e-6)
