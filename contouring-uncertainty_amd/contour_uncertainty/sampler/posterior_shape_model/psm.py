"""``PosteriorShapeModelSampler`` on the MI355X kernel (reference sampler/posterior_shape_model/psm.py:23-450).

Same constructor (``psm_path``, ``levels``), ``get_points_order``, ``__call__(mu, cov, n=...) -> (n, K, 2)``,
``transform`` / ``inverse_transform`` / ``merge_priors``.  ``sample_batch`` is the batched entry the tasks use: all
(frame, t_e) pairs and all n samples in ONE kernel launch (frames are independent -> frame-sharded on N GPUs).
The reference's per-frame ``torch.linalg.eig`` PCA and per-sample 42x42 inverses are replaced by closed forms that
depend on the frame and level only (see csrc/psm_sampler.hip).
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional

import numpy as np
import torch

from contour_uncertainty.sampler.posterior_shape_model.utils import index_to_flat
from contour_uncertainty.sampler.sampler import Sampler
from cu_hip import ops

TABLE_STRIDE = 2 + 48 + 32


def load_psm(path) -> dict:
    """The shipped PSM files are pickled dicts in .npy (reference psm.py:32-38); plain .npz copies load without pickle."""
    path = Path(path)
    if path.suffix == ".npz" or not path.exists() and path.with_suffix(".npz").exists():
        return dict(np.load(path if path.suffix == ".npz" else path.with_suffix(".npz")))
    return np.load(str(path), allow_pickle=True).item()


class PosteriorShapeModelSampler(Sampler):
    def __init__(self, psm_path: Path, levels: int = 3):
        data = load_psm(psm_path)
        self.mu = torch.tensor(np.asarray(data["mu"]), dtype=torch.float)
        self.Q = torch.tensor(np.asarray(data["Q"]), dtype=torch.float)
        self.mean = torch.tensor(np.asarray(data["scaler_mean"]), dtype=torch.float)
        self.scale = torch.tensor(np.asarray(data["scaler_scale"]), dtype=torch.float)
        self.X_train = torch.tensor(np.asarray(data["X_train"]), dtype=torch.float)
        self.X_val = torch.tensor(np.asarray(data["X_val"]), dtype=torch.float)
        self.nb_points = self.mu.shape[0] // 2
        self.initial_points, self.points_order = self.get_points_order(self.nb_points, levels=levels)
        # frame-independent part of the PCA covariance: C_f = Cov0 + (xbar - m_f)(xbar - m_f)^T
        x = self.X_train.double()
        self._xbar = x.mean(0)
        self._cov0 = ((x - self._xbar).T @ (x - self._xbar) / x.shape[0]).float()
        self._xbar = self._xbar.float()
        self._dev_cache = {}
        self._build_tables()

    def _build_tables(self):
        """Per level: flat indices already known (sorted) and the points produced (psm.py:262-368)."""
        k = self.nb_points
        known = sorted(self.initial_points)
        rows, sigma2, sample = [], [], []
        for pts in self.points_order:
            if len(known) == k:
                break
            rows.append((index_to_flat(known), list(pts)))
            sigma2.append(1.0)          # sigmas = [1, 1, 1, 1]  (psm.py:216)
            sample.append(1)
            known = sorted(known + list(pts))
        rest = [j for j in range(k) if j not in known]
        if rest:                        # complete_shape: PSM mean with slack 0.001 (psm.py:357-368)
            rows.append((index_to_flat(known), rest))
            sigma2.append(0.001)
            sample.append(0)
        tab = np.zeros((len(rows), TABLE_STRIDE), dtype=np.int32)
        for i, (g, t) in enumerate(rows):
            tab[i, 0], tab[i, 1] = len(g), len(t)
            tab[i, 2:2 + len(g)] = g
            tab[i, 2 + 48:2 + 48 + len(t)] = t
        self._tables, self._sigma2, self._sample = torch.from_numpy(tab), sigma2, sample
        self._rec_stride = ops.psm_record_floats([len(g) for g, _ in rows], [len(t) for _, t in rows])

    def _on(self, device):
        key = str(device)
        if key not in self._dev_cache:
            self._dev_cache[key] = tuple(t.to(device).contiguous() for t in
                                         (self._cov0, self._xbar, self.mean, self.scale, self._tables))
        return self._dev_cache[key]

    def sample_batch(self, mu: torch.Tensor, cov: torch.Tensor, n: int = 1, eps: Optional[torch.Tensor] = None,
                     seed: Optional[int] = None) -> torch.Tensor:
        """mu (F, K, 2), cov (F, K, 2, 2) -> (F, n, K, 2) on the GPU."""
        dev = torch.device("cuda", torch.cuda.current_device()) if not mu.is_cuda else mu.device
        mu = mu.to(dev, torch.float32).contiguous()
        cov = cov.to(dev, torch.float32)
        cov3 = torch.stack([cov[..., 0, 0], cov[..., 1, 1], cov[..., 1, 0]], -1).contiguous()
        cov0, xbar, smean, sscale, tables = self._on(dev)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if eps is not None:
            eps = eps.to(dev, torch.float32).contiguous()
        return ops.psm_sample_gauss(mu, cov3, cov0, xbar, smean, sscale, self.initial_points, tables, self._sigma2,
                                    self._sample, n, eps, seed)

    def sample_batch_skew(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor, n: int = 1, skew_bits: int = 0,
                          eps: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None,
                          seed: Optional[int] = None, prior_mu: Optional[torch.Tensor] = None,
                          prior_cov: Optional[torch.Tensor] = None, use_initial_pdf: bool = False,
                          flip_alpha_y: bool = True, grid_size: int = 256) -> torch.Tensor:
        """mu (F,K,2), cov (F,K,2,2), alpha (F,K,2) -> (F, n, K, 2): anchors by ``rvs_fast`` (alpha_y negated first,
        psm.py:235-236 / psm_skew.py:232), points whose bit is set in ``skew_bits`` by the grid product, the others by
        the product-of-Gaussians merge (``cu_psm_setup`` + ``cu_psm_sample_skew``).

        eps (F,n,K,3) standard normals / u (F,n,K) uniforms in [0,1) replace the internal generator (parity tests).
        prior_mu (F,n,K,2) / prior_cov (F,n,K,2,2): extra Gaussian factor of every point's table (the ED/ES coupling of
        SequenceSkewPSMSampler); with ``use_initial_pdf`` the anchors are drawn from skew-pdf x prior as well."""
        dev = torch.device("cuda", torch.cuda.current_device()) if not mu.is_cuda else mu.device
        f32 = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()
        lower3 = lambda c: torch.stack([c[..., 0, 0], c[..., 1, 1], c[..., 1, 0]], -1).contiguous()
        mu, alpha = f32(mu), f32(alpha)
        cov3 = lower3(cov.to(dev, torch.float32))
        cov0, xbar, smean, sscale, tables = self._on(dev)
        rec = ops.psm_setup(mu.reshape(mu.shape[0], -1), cov0, xbar, smean, sscale, tables, self._sigma2, self._rec_stride)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        p3 = None if prior_cov is None else lower3(prior_cov.to(dev, torch.float32))
        return ops.psm_sample_skew(mu, cov3, alpha, -1.0 if flip_alpha_y else 1.0, skew_bits, rec, smean, sscale,
                                   self.initial_points, tables, self._sample, n, f32(prior_mu), p3, use_initial_pdf,
                                   grid_size, f32(eps), f32(u), seed)

    def __call__(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor = None, n: int = 1,
                 debug_img=None) -> torch.Tensor:
        """mu (K, 2), cov (K, 2, 2) -> (n, K, 2) on mu's device (reference psm.py:73-93).  With ``alpha`` the anchors
        are skew-normal draws and every other point stays Gaussian (psm.py:233-238)."""
        if alpha is not None:
            return self.sample_batch_skew(mu[None], cov[None], alpha[None], n=n, skew_bits=0)[0].to(mu.device)
        return self.sample_batch(mu[None], cov[None], n=n)[0].to(mu.device)

    @staticmethod
    def merge_priors(mu1, cov1, mu2, cov2, p: float = 0.5):
        """Product-of-Gaussians merge (reference psm.py:424-440; ``p`` is ignored there too)."""
        w = torch.inverse(cov1 + cov2)
        return cov1 @ w @ mu2[..., None] + cov2 @ w @ mu1[..., None], cov1 @ w @ cov2

    def transform(self, s):
        return ((s.reshape(1, -1) - self.mean.to(s.device)) / self.scale.to(s.device)).reshape(s.shape)

    def inverse_transform(self, s):
        return ((s.reshape(1, -1) * self.scale.to(s.device)) + self.mean.to(s.device)).reshape(s.shape)
