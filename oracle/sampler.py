"""Oracle (test infrastructure): the Gaussian posterior-shape-model contour sampler, restated from the reference's text.

  * ``pca``                    reference contour_uncertainty/sampler/posterior_shape_model/posteriorshapemodel.py:9-46
  * ``posterior_shape_model``  reference .../posteriorshapemodel.py:49-81 (row masks, sigma2 slack)
  * ``index_to_flat``          reference .../posterior_shape_model/utils.py:4-25
  * ``get_points_order``       reference .../posterior_shape_model/psm.py:43-71
  * ``merge_priors``           reference .../psm.py:424-440 (ignores its ``p`` argument)
  * ``sample_endo_contour``    reference .../psm.py:199-384 (Gaussian branch, ``complete_shape=True``, no debug plots)
  * ``sample_points``          reference .../psm.py:387-421: ``MultivariateNormal(mu, cov).rsample`` = mu + chol(cov) @ eps

  * ``numerical_sampling`` / ``SkewPosteriorShapeModelSampler``  reference .../psm_skew.py:45-158, 162-503
  * ``SequencePSMSampler``     reference .../sequence_sampler.py:13-160
  * ``SequenceSkewPSMSampler`` reference .../psm_skew_sequence.py:21-166
  * ``rvs_fast``               reference contour_uncertainty/distributions/bivariateskewnormal.py:159-191

Randomness of the grid sampler: the reference draws the cell with ``torch.multinomial``; any exact categorical draw from
the same normalised table is equivalent, and the oracle uses the inverse CDF of a supplied uniform ``u`` (flat x-major
order of ``meshgrid(indexing='ij')``) so that an implementation can be compared draw by draw.  The tables themselves
are pinned by ``tests/golden/skew_grid.npz`` (outputs of the importable pdf classes).

psm.py itself cannot be imported (it needs the missing module ``contour_uncertainty.data.ultromics``); its
deterministic building blocks are pinned by ``tests/golden/psm_math.npz`` (outputs of the importable ``pca`` /
``posterior_shape_model`` / ``get_points_order``).  To compare a sampler implementation exactly, the standard-normal
draws ``eps`` can be supplied (``eps[k]`` is used when point k is sampled).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


def index_to_flat(indices) -> List[int]:
    if isinstance(indices, int):
        return [indices * 2, indices * 2 + 1]
    out: List[int] = []
    for i in indices:
        out.extend([i * 2, i * 2 + 1])
    return out


def get_points_order(nb_points: int = 21, nb_initial_points: int = 3, levels: Optional[int] = None):
    initial = np.round(np.linspace(0, nb_points - 1, nb_initial_points)).astype(int).tolist()
    levels = levels or int(math.log(nb_points, 2))
    all_points, order = list(initial), []
    for _ in range(levels):
        lvl = []
        for j in range(len(all_points) - 1):
            if all_points[j] + 1 != all_points[j + 1]:
                p = (all_points[j] + all_points[j + 1]) / 2
                p = math.ceil(p) if p > nb_points / 2 else math.floor(p)
                lvl.append(int(p))
        if not lvl:
            break
        all_points.extend(lvl)
        all_points.sort()
        order.append(lvl)
    return initial, order


def pca(X: Tensor, mu: Optional[Tensor] = None):
    Xc = X[..., None]
    mu = mu if mu is not None else Xc.mean(axis=0)
    diff = Xc.squeeze().T - mu
    cov = torch.einsum("ij,kj", diff, diff) / Xc.shape[0]
    vals, vecs = torch.linalg.eig(cov)
    vals, vecs = vals.real.abs(), vecs.real
    idx = vals.argsort().flip(0)
    vals, vecs = vals[idx], vecs[:, idx]
    return mu, torch.mm(vecs, torch.diag(torch.sqrt(vals)))


def posterior_shape_model(s_g: Tensor, g_indices: Sequence[int], mu: Tensor, Q: Tensor, sigma2: float = 1):
    p = len(mu)
    eye = torch.eye(p)
    mu_mask = torch.zeros(p, 1)
    mu_mask[list(g_indices)] = 1
    q_mask = torch.zeros(p, p)
    q_mask[list(g_indices)] = 1
    mu_g, Q_g, s_g = mu * mu_mask, Q * q_mask, s_g * mu_mask
    inv = torch.inverse(Q_g.T @ Q_g + sigma2 * eye)
    mu_c = mu + Q @ inv @ Q_g.T @ (s_g - mu_g)
    cov_c = sigma2 * Q @ inv @ Q.T
    return mu_c, cov_c


def merge_priors(mu1: Tensor, cov1: Tensor, mu2: Tensor, cov2: Tensor):
    w = torch.inverse(cov1 + cov2)
    sigma_f = cov1 @ w @ cov2
    mu_f = cov1 @ w @ mu2[..., None] + cov2 @ w @ mu1[..., None]
    return mu_f, sigma_f


class GaussianPSMSamplerOracle:
    def __init__(self, psm: dict, levels: int = 3, dtype=torch.float):
        """dtype=torch.float is the reference's arithmetic; float64 (with torch.set_default_dtype) gives the same
        algorithm without its rounding noise."""
        self.mean = torch.as_tensor(np.asarray(psm["scaler_mean"]), dtype=dtype)
        self.scale = torch.as_tensor(np.asarray(psm["scaler_scale"]), dtype=dtype)
        self.X_train = torch.as_tensor(np.asarray(psm["X_train"]), dtype=dtype)
        k = self.X_train.shape[1] // 2
        self.initial_points, self.points_order = get_points_order(k, levels=levels)

    def transform(self, s):
        return ((s.reshape(1, -1) - self.mean) / self.scale).reshape(s.shape)

    def inverse_transform(self, s):
        return ((s.reshape(1, -1) * self.scale) + self.mean).reshape(s.shape)

    @staticmethod
    def _draw(mu_j, cov_j, eps_j):
        return mu_j + torch.linalg.cholesky(cov_j) @ eps_j

    def sample_one(self, mu_p: Tensor, cov_p: Tensor, pca_mu: Tensor, Q: Tensor, eps: Tensor) -> Tensor:
        """One contour (K,2) given the K x 2 standard-normal draws ``eps`` (psm.py:199-384)."""
        k = mu_p.shape[0]
        contour = torch.zeros_like(mu_p)
        sampled = list(self.initial_points)
        for j in self.initial_points:
            contour[j] = self._draw(mu_p[j], cov_p[j], eps[j])
        sigmas = [1, 1, 1, 1]
        for i, points in enumerate(self.points_order):
            sampled.sort()
            if len(sampled) == k:
                break
            s_g = self.transform(contour).reshape(-1, 1)
            mu_c, cov_c = posterior_shape_model(s_g, index_to_flat(sampled), pca_mu, Q, sigma2=sigmas[i])
            mu_c = self.inverse_transform(mu_c.squeeze()).reshape(mu_p.shape)
            cov_c = cov_c * self.scale
            cov_c = torch.stack([cov_c[2 * j:2 * j + 2, 2 * j:2 * j + 2] for j in range(k)])
            mu_f, cov_f = merge_priors(mu_p, cov_p, mu_c, cov_c)
            mu_f = mu_f.squeeze(-1)
            for j in points:
                contour[j] = self._draw(mu_f[j], cov_f[j], eps[j])
            sampled.extend(points)
        sampled.sort()
        if len(sampled) != k:
            s_g = self.transform(contour).reshape(-1, 1)
            mu_c, _ = posterior_shape_model(s_g, index_to_flat(sampled), pca_mu, Q, sigma2=0.001)
            mu_c = self.inverse_transform(mu_c.squeeze()).reshape(mu_p.shape)
            rest = [j for j in range(k) if j not in sampled]
            contour[rest] = mu_c[rest]
        return contour

    def __call__(self, mu: Tensor, cov: Tensor, n: int = 1, eps: Optional[Tensor] = None,
                 generator: Optional[torch.Generator] = None) -> Tensor:
        """mu (K,2), cov (K,2,2) -> (n,K,2)  (psm.py:73-93: PCA re-centred on the predicted contour)."""
        pca_mu, Q = pca(self.X_train, self.transform(mu).reshape(-1, 1))
        if eps is None:
            eps = torch.randn(n, mu.shape[0], 2, generator=generator)
        return torch.stack([self.sample_one(mu, cov, pca_mu, Q, eps[i]) for i in range(n)])


# ---------------------------------------------------------------------------------------------------------------------
# skew-normal grid sampler
# ---------------------------------------------------------------------------------------------------------------------
def _batch_matrix_pow(matrix: Tensor, p: float) -> Tensor:
    """distributions/utils.py:100-129 (eig based)."""
    vals, vecs = torch.linalg.eig(matrix)
    vals_pow = vals.contiguous().pow(p).real
    vecs = vecs.real
    return torch.matmul(vecs, torch.matmul(torch.diag_embed(vals_pow), torch.inverse(vecs)))


def gauss_logpdf(x: Tensor, loc: Tensor, cov: Tensor) -> Tensor:
    """BivariateNormal.logpdf (bivariatenormal.py:15-32) for ONE distribution on points x (..., 2)."""
    shape = x.shape[:-1]
    d = (x.reshape(-1, 2) - loc)[..., None]
    t3 = (d.transpose(-1, -2) @ torch.inverse(cov)) @ d
    lp = -math.log(2 * math.pi) - torch.log(torch.det(cov)) / 2 - t3.squeeze() / 2
    return lp.reshape(shape)


def skew_logpdf(x: Tensor, loc: Tensor, cov: Tensor, alpha: Tensor) -> Tensor:
    """BivariateSkewNormal.logpdf (bivariateskewnormal.py:19-48) for ONE distribution."""
    shape = x.shape[:-1]
    d = (x.reshape(-1, 2) - loc)[..., None]
    z = (alpha[None, None, :] @ _batch_matrix_pow(cov[None], -0.5)) @ d
    cdf = 0.5 * (1 + torch.erf(z.squeeze() / math.sqrt(2)))
    return (math.log(2) + gauss_logpdf(x, loc, cov).reshape(-1) + torch.log(cdf + 1e-7)).reshape(shape)


def mvn_pdf(x: Tensor, loc: Tensor, cov: Tensor) -> Tensor:
    """exp(MultivariateNormal(loc, cov).log_prob(x)) (psm_skew.py:78)."""
    return torch.exp(torch.distributions.MultivariateNormal(loc, cov, validate_args=False).log_prob(x))


def make_grid(grid_size: int = 256):
    x = torch.linspace(0, 255, grid_size)
    X, Y = torch.meshgrid(x, x, indexing="ij")
    return X, Y, torch.stack([X, Y], dim=-1)


def inverse_cdf_pick(p: Tensor, u: float) -> int:
    """Index of the cell a uniform u in [0,1) selects in the (unnormalised, non-negative) table p, flat order."""
    c = torch.cumsum(p.flatten().double(), 0)
    idx = int(torch.searchsorted(c, torch.tensor(float(u) * float(c[-1]), dtype=torch.double), right=True))
    return min(idx, p.numel() - 1)


def numerical_sampling(p1: Tensor, mu2: Tensor, cov2: Tensor, X: Tensor, Y: Tensor, grid: Tensor, u: float) -> Tensor:
    """psm_skew.py:45-158 with p1 supplied (as the sampler always does, :319-326)."""
    p = p1 * mvn_pdf(grid, mu2, cov2)
    tot = torch.sum(p)
    if not (torch.isfinite(tot) and tot > 0):          # torch.multinomial raises -> `except:` returns mu2 (:135-154)
        return mu2.clone()
    p = p / tot
    idx = inverse_cdf_pick(p, u)
    return torch.stack([X.flatten()[idx], Y.flatten()[idx]])


def rvs_fast(mu: Tensor, cov: Tensor, alpha: Tensor, eps3: Tensor) -> Tensor:
    """bivariateskewnormal.py:159-191 with the standard normals of MultivariateNormal.sample supplied (eps3 (..., 3))."""
    delta = (1 / torch.sqrt(1 + alpha @ cov @ alpha)) * cov @ alpha
    cs = torch.zeros(3, 3, dtype=mu.dtype)
    cs[0, 0] = 1
    cs[1:, 0] = delta
    cs[0, 1:] = delta
    cs[1:, 1:] = cov
    x = eps3.reshape(-1, 3) @ torch.linalg.cholesky(cs).T
    x0, x1 = x[:, 0], x[:, 1:].clone()
    x1[x0 <= 0] = -x1[x0 <= 0]
    return (x1 + mu[None]).reshape(eps3.shape[:-1] + (2,))


class SkewPSMSamplerOracle(GaussianPSMSamplerOracle):
    """SkewPosteriorShapeModelSampler (psm_skew.py:162-503), no debug plots.  Points outside ``skew_indices`` call the
    undefined ``merge_gaussian_priors`` in the reference (:329, an AttributeError); the product-of-Gaussians
    ``merge_priors`` of the parent sampler is what the name and the call signature say, and is used here."""

    def __init__(self, psm: dict, levels: int = 3, skew_indices=None, grid_size: int = 256, dtype=torch.float):
        """dtype=float64 (together with torch.set_default_dtype(torch.float64)) runs the same algorithm without the
        reference's f32 rounding noise in the PSM algebra (a few tenths of a pixel at the deepest level)."""
        super().__init__(psm, levels, dtype=dtype)
        k = self.X_train.shape[1] // 2
        self.skew_indices = list(range(k)) if skew_indices is None else list(skew_indices)
        self.X, self.Y, self.grid_points = make_grid(grid_size)

    def compute_psm(self, contour, sampled, sigma, pca_mu, Q):
        s_g = self.transform(contour).reshape(-1, 1)
        mu_c, cov_c = posterior_shape_model(s_g, index_to_flat(sampled), pca_mu, Q, sigma2=sigma)
        mu_c = self.inverse_transform(mu_c.squeeze()).reshape(contour.shape)
        cov_c = cov_c * self.scale
        return mu_c, torch.stack([cov_c[2 * j:2 * j + 2, 2 * j:2 * j + 2] for j in range(contour.shape[0])])

    def sample_contour(self, mu_p, cov_p, alpha_p, pdfs, pca_mu, Q, eps3, u, use_initial_pdf=False):
        """psm_skew.py:247-411.  eps3 (K,3): normals of the anchors' rvs_fast / the Gaussian points; u (K,): uniforms."""
        k = mu_p.shape[0]
        contour = torch.zeros_like(mu_p)
        sampled = list(self.initial_points)
        for j in self.initial_points:
            if use_initial_pdf:
                idx = inverse_cdf_pick(pdfs[j], float(u[j]))
                contour[j] = torch.stack([self.X.flatten()[idx], self.Y.flatten()[idx]])
            else:
                contour[j] = rvs_fast(mu_p[j], cov_p[j], alpha_p[j], eps3[j])
        sigmas = [1, 1, 1, 1]
        for i, points in enumerate(self.points_order):
            sampled.sort()
            if len(sampled) == k:
                break
            mu_c, cov_c = self.compute_psm(contour, sampled, sigmas[i], pca_mu, Q)
            new = {}
            for j in points:
                if j in self.skew_indices:
                    new[j] = numerical_sampling(pdfs[j], mu_c[j], cov_c[j], self.X, self.Y, self.grid_points, float(u[j]))
                else:
                    mu_f, cov_f = merge_priors(mu_p, cov_p, mu_c, cov_c)
                    new[j] = self._draw(mu_f.squeeze(-1)[j], cov_f[j], eps3[j, :2])
            for j, v in new.items():
                contour[j] = v
            sampled.extend(points)
        sampled.sort()
        if len(sampled) != k:
            s_g = self.transform(contour).reshape(-1, 1)
            mu_c, _ = posterior_shape_model(s_g, index_to_flat(sampled), pca_mu, Q, sigma2=0.001)
            mu_c = self.inverse_transform(mu_c.squeeze()).reshape(mu_p.shape)
            rest = [j for j in range(k) if j not in sampled]
            contour[rest] = mu_c[rest]
        return contour

    def sample_one_instant(self, mu, cov, alpha, n, eps3, u, pdfs=None, use_initial_pdf=False):
        """psm_skew.py:210-244: PCA about the prediction, alpha_y negated (:232), pdf tables once per frame."""
        pca_mu, Q = pca(self.X_train, self.transform(mu).reshape(-1, 1))
        alpha = alpha.clone() * torch.tensor([1.0, -1.0])
        if pdfs is None:
            pdfs = torch.stack([torch.exp(skew_logpdf(self.grid_points, mu[i], cov[i], alpha[i]))
                                for i in range(mu.shape[0])])
        return torch.stack([self.sample_contour(mu, cov, alpha, pdfs, pca_mu, Q, eps3[i], u[i], use_initial_pdf)
                            for i in range(n)])

    def __call__(self, mu, cov, alpha, n, eps3, u):
        """mu (B,K,2) ... eps3 (B,n,K,3), u (B,n,K) -> (B,n,K,2)  (psm_skew.py:187-208)."""
        return torch.stack([self.sample_one_instant(mu[b], cov[b], alpha[b], n, eps3[b], u[b]) for b in range(mu.shape[0])])


class _SequenceMixin:
    def _load_seq(self, seq: dict, dtype=torch.float):
        f = lambda k: torch.as_tensor(np.asarray(seq[k]), dtype=dtype)
        self.seq_mu, self.seq_Q = f("mu"), f("Q")
        self.seq_mean, self.seq_scale, self.seq_X_train = f("scaler_mean"), f("scaler_scale"), f("X_train")

    def sequence_transform(self, s):
        return ((s.reshape(1, -1) - self.seq_mean) / self.seq_scale).reshape(s.shape)

    def sequence_inverse_transform(self, s):
        return ((s.reshape(1, -1) * self.seq_scale) + self.seq_mean).reshape(s.shape)

    def _second_instant_model(self, s_first, first, mu_shape, seq_mu, seq_Q):
        """sequence_sampler.py:72-86 / psm_skew_sequence.py:72-84."""
        k = mu_shape[1]
        s_g = torch.zeros(mu_shape)
        s_g[first] = s_first
        s_g = self.sequence_transform(s_g).reshape(-1, 1)
        idx = list(range(k)) if first == 0 else list(range(k, 2 * k))
        mu_c, cov_c = posterior_shape_model(s_g, index_to_flat(idx), seq_mu, seq_Q, sigma2=1)
        mu_c = self.sequence_inverse_transform(mu_c.squeeze()).reshape((2 * k, 2))
        cov_c = cov_c * self.seq_scale
        cov_c = torch.stack([cov_c[2 * i:2 * i + 2, 2 * i:2 * i + 2] for i in range(2 * k)])
        return mu_c, cov_c


class SequencePSMSamplerOracle(GaussianPSMSamplerOracle, _SequenceMixin):
    """sequence_sampler.py:13-160 (alpha is never forwarded by ``__call__``, :48)."""

    def __init__(self, psm: dict, seq: dict, levels: int = 3, dtype=torch.float):
        super().__init__(psm, levels, dtype=dtype)
        self._load_seq(seq, dtype)

    def sample_two_contours(self, mu, cov, first, eps):
        """mu (2,K,2), cov (2,K,2,2), eps (2,K,2) (row i = draws for instant i) -> dict like the reference's."""
        second = 1 - first
        k = mu.shape[1]
        s = torch.zeros_like(mu)
        s[first] = super().__call__(mu[first], cov[first], 1, eps=eps[first][None])[0]
        mu_c, cov_c = self._second_instant_model(s[first].clone(), first, mu.shape, self.seq_mu, self.seq_Q)
        mu_f, cov_f = merge_priors(mu.reshape(2 * k, 2), cov.reshape(2 * k, 2, 2), mu_c.to(mu.dtype), cov_c.to(mu.dtype))
        mu_f, cov_f = mu_f.reshape(2, k, 2), cov_f.reshape(2, k, 2, 2)
        s[second] = super().__call__(mu_f[second], cov_f[second], 1, eps=eps[second][None])[0]
        return {"mu_c": mu_c.reshape(2, k, 2), "cov_c": cov_c.reshape(2, k, 2, 2), "mu_f": mu_f, "cov_f": cov_f, "s": s}

    def sample(self, mu, cov, firsts, eps):
        """firsts: the n values random.randint(0,1) returned (sequence_sampler.py:47); eps (n,2,K,2) -> (n,2,K,2)."""
        return torch.stack([self.sample_two_contours(mu, cov, f, eps[i])["s"] for i, f in enumerate(firsts)])


class SequenceSkewPSMSamplerOracle(SkewPSMSamplerOracle, _SequenceMixin):
    """psm_skew_sequence.py:21-166.  Quirk kept: the second instant's skew tables use alpha as given (:90), while the
    first instant's use alpha_y negated (sample_one_instant, psm_skew.py:232)."""

    def __init__(self, psm: dict, seq: dict, levels: int = 3, skew_indices=None, dtype=torch.float):
        super().__init__(psm, levels, skew_indices, dtype=dtype)
        self._load_seq(seq, dtype)

    def sample_two_contours(self, mu, cov, alpha, first, eps3, u):
        """eps3 (2,K,3), u (2,K): row i = draws for instant i."""
        second = 1 - first
        k = mu.shape[1]
        seq_mu, seq_Q = pca(self.seq_X_train, self.sequence_transform(mu).reshape(-1, 1))
        s = torch.zeros_like(mu)
        s[first] = self.sample_one_instant(mu[first], cov[first], alpha[first], 1, eps3[first][None], u[first][None])[0]
        mu_c, cov_c = self._second_instant_model(s[first].clone(), first, mu.shape, seq_mu, seq_Q)
        mu_c, cov_c = mu_c.reshape(2, k, 2), cov_c.reshape(2, k, 2, 2)
        pdfs = []
        for i in range(k):
            p = torch.exp(skew_logpdf(self.grid_points, mu[second, i], cov[second, i], alpha[second, i])) * \
                torch.exp(gauss_logpdf(self.grid_points, mu_c[second, i], cov_c[second, i]))
            pdfs.append(p / p.sum())
        s[second] = self.sample_one_instant(mu[second], cov[second], alpha[second], 1, eps3[second][None], u[second][None],
                                            pdfs=torch.stack(pdfs), use_initial_pdf=True)[0]
        return s, mu_c, cov_c

    def sample(self, mu, cov, alpha, firsts, eps3, u):
        """-> (2, n, K, 2) like psm_skew_sequence.py:48."""
        out = torch.stack([self.sample_two_contours(mu, cov, alpha, f, eps3[i], u[i])[0] for i, f in enumerate(firsts)])
        return out.permute(1, 0, 2, 3)
