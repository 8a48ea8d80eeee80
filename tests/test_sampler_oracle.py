"""Sampler oracle (CPU restatement of psm.py / posteriorshapemodel.py) vs golden vectors generated from the reference's
importable ``pca`` / ``posterior_shape_model`` / ``get_points_order``."""
import numpy as np
import torch

from oracle import sampler as S


def test_points_order_matches_reference(golden_dir):
    g = np.load(golden_dir / "psm_math.npz")
    init, order = S.get_points_order(21, levels=3)
    assert init == g["order_init"].tolist() == [0, 10, 20]
    assert order == [g["order_l0"].tolist(), g["order_l1"].tolist(), g["order_l2"].tolist()]
    assert order == [[5, 15], [2, 7, 13, 18], [1, 3, 6, 8, 12, 14, 17, 19]]


def test_pca_and_posterior_shape_model_match_reference(golden_dir):
    g = np.load(golden_dir / "psm_math.npz")
    psm = np.load(golden_dir / "camus-cont_psm_11_no_std.npz")
    X = torch.tensor(psm["X_train"]).float()
    mu_pred = torch.tensor(g["mu_pred"])
    mu_p, Q = S.pca(X, mu_pred)
    assert torch.allclose(Q @ Q.T, torch.tensor(g["pca_QQt"]), rtol=1e-3, atol=1e-2)
    s = torch.tensor(g["psm_s"])
    init, order = S.get_points_order(21, levels=3)
    known = list(init)
    for i, lv in enumerate(order):
        idx = S.index_to_flat(sorted(known))
        assert idx == g[f"psm_l{i}_idx"].tolist()
        mu_c, cov_c = S.posterior_shape_model(s, idx, mu_pred, Q, sigma2=1)
        assert torch.allclose(mu_c, torch.tensor(g[f"psm_l{i}_mu"]), rtol=1e-3, atol=2e-2)
        assert torch.allclose(cov_c, torch.tensor(g[f"psm_l{i}_cov"]), rtol=1e-3, atol=2e-2)
        known += lv
    mu_c, _ = S.posterior_shape_model(s, S.index_to_flat(sorted(known)), mu_pred, Q, sigma2=0.001)
    assert torch.allclose(mu_c, torch.tensor(g["psm_final_mu"]), rtol=1e-3, atol=5e-2)


def test_conditional_is_woodbury_form(golden_dir):
    """The identity the GPU sampler relies on: with C = Q Q^T,
    mu_c = mu + C[:,g] (C[g,g] + s2 I)^-1 (s_g - mu_g),  cov_c = C - C[:,g] (C[g,g] + s2 I)^-1 C[g,:]."""
    psm = np.load(golden_dir / "camus-cont_psm_11_no_std.npz")
    X = torch.tensor(psm["X_train"]).double()
    mu = X[7].reshape(-1, 1) + 1.5
    _, Q = S.pca(X.float(), mu.float())
    Q = Q.double()
    C = Q @ Q.T
    # C is also the population covariance about the training mean plus a rank-1 term
    xbar = X.mean(0, keepdim=True).T
    C2 = (X.T - xbar) @ (X.T - xbar).T / X.shape[0] + (xbar - mu) @ (xbar - mu).T
    assert torch.allclose(C, C2, rtol=1e-3, atol=1e-2)
    g = S.index_to_flat([0, 5, 10, 15, 20])
    s = mu + torch.randn(42, 1, dtype=torch.double)
    mu_c, cov_c = S.posterior_shape_model(s.float(), g, mu.float(), Q.float(), sigma2=1)
    A = torch.inverse(C[g][:, g] + torch.eye(len(g), dtype=torch.double))
    G = C[:, g] @ A
    assert torch.allclose(mu_c.double(), mu + G @ (s - mu)[g], rtol=1e-3, atol=1e-2)
    assert torch.allclose(cov_c.double(), C - G @ C[g, :], rtol=1e-3, atol=1e-2)


def test_sampler_statistics_anchor_points(golden_dir):
    psm = dict(np.load(golden_dir / "camus-cont_psm_11_no_std.npz"))
    orc = S.GaussianPSMSamplerOracle(psm)
    mu = torch.tensor(psm["X_val"][3] + psm["scaler_mean"]).float().reshape(21, 2)
    cov = torch.eye(2)[None].repeat(21, 1, 1) * 9.0
    cov[:, 0, 1] = cov[:, 1, 0] = 2.0
    out = orc(mu, cov, n=400, generator=torch.Generator().manual_seed(1))
    assert out.shape == (400, 21, 2)
    for j in (0, 10, 20):       # anchors are drawn from the predicted distribution itself
        assert torch.allclose(out[:, j].mean(0), mu[j], atol=0.6)
        c = torch.cov(out[:, j].T)
        assert torch.allclose(c, cov[j], atol=2.5)


def test_skew_grid_tables_match_reference(golden_dir):
    """oracle densities / normalised product / rvs_fast / two-instant conditional vs outputs of the imported reference."""
    from oracle import sampler as S
    g = np.load(golden_dir / "skew_grid.npz")
    X, Y, grid = S.make_grid(256)
    for i in range(3):
        m1, c1, a1, m2, c2 = (torch.from_numpy(g[f"c{i}_{k}"]) for k in ("mu1", "cov1", "alpha1", "mu2", "cov2"))
        p1 = torch.exp(S.skew_logpdf(grid, m1, c1, a1))
        p2 = S.mvn_pdf(grid, m2, c2)
        assert torch.allclose(p1, torch.from_numpy(g[f"c{i}_p1"]), rtol=2e-4, atol=1e-9)
        assert torch.allclose(p2, torch.from_numpy(g[f"c{i}_p2"]), rtol=2e-4, atol=1e-12)
        assert torch.allclose(torch.exp(S.gauss_logpdf(grid, m2, c2)), torch.from_numpy(g[f"c{i}_p2_bn"]), rtol=2e-4, atol=1e-12)
        p = p1 * p2
        p = p / p.sum()
        ref = torch.from_numpy(g[f"c{i}_p"])
        assert torch.allclose(p, ref, rtol=5e-4, atol=1e-10)
        # the inverse-CDF pick selects the same cells from the oracle's and the reference's table
        for u in (0.0, 0.013, 0.25, 0.5, 0.77, 0.999):
            a, b = S.inverse_cdf_pick(p, u), S.inverse_cdf_pick(ref, u)
            assert abs(a - b) <= 1 or abs(abs(a - b) - 256) <= 1
    x = S.rvs_fast(torch.from_numpy(g["rvs_mu"]), torch.from_numpy(g["rvs_cov"]), torch.from_numpy(g["rvs_alpha"]),
                   torch.from_numpy(g["rvs_eps"]))
    assert torch.allclose(x, torch.from_numpy(g["rvs_x"]), rtol=1e-5, atol=1e-4)


def test_sequence_conditional_matches_reference(golden_dir):
    """Two-instant PSM conditional (sequence_sampler.py:83, psm_skew_sequence.py:66,78) on the shipped 84-dim model."""
    from oracle import sampler as S
    g = np.load(golden_dir / "skew_grid.npz")
    seq = dict(np.load(golden_dir / "camus-cont_sequence_psm_11_no_std.npz"))
    smu, sQ = torch.from_numpy(seq["mu"]).float(), torch.from_numpy(seq["Q"]).float()
    blocks = lambda c: torch.stack([c[2 * i:2 * i + 2, 2 * i:2 * i + 2] for i in range(42)])
    for first in (0, 1):
        sg = torch.from_numpy(g[f"seq{first}_sg"]).reshape(-1, 1)
        idx = S.index_to_flat(list(range(21)) if first == 0 else list(range(21, 42)))
        mu_c, cov_c = S.posterior_shape_model(sg, idx, smu, sQ, sigma2=1)
        assert torch.allclose(mu_c.squeeze(), torch.from_numpy(g[f"seq{first}_mu_c"]), rtol=1e-4, atol=5e-3)
        assert torch.allclose(blocks(cov_c), torch.from_numpy(g[f"seq{first}_cov_blocks"]), rtol=1e-3, atol=5e-3)
        pm, pQ = S.pca(torch.from_numpy(seq["X_train"]).float(), torch.from_numpy(g["seq_xv"]).reshape(-1, 1))
        mu_c, cov_c = S.posterior_shape_model(sg, idx, pm, pQ, sigma2=1)
        assert torch.allclose(mu_c.squeeze(), torch.from_numpy(g[f"seqpca{first}_mu_c"]), rtol=1e-3, atol=5e-2)
        assert torch.allclose(blocks(cov_c), torch.from_numpy(g[f"seqpca{first}_cov_blocks"]), rtol=1e-2, atol=5e-2)
