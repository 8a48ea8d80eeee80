"""``SequencePSMSampler`` on the MI355X kernels (reference sampler/posterior_shape_model/sequence_sampler.py:13-160).

ED/ES pair: one instant (picked with ``random.randint`` like the reference, :47) is sampled with the Gaussian PSM
sampler, the fixed 84-dimensional two-instant PSM is conditioned on it (``cu_psm_condition``; its gains do not depend
on the data and are computed once, in f64, when the model file is loaded), merged with the other instant's prediction
and that instant is sampled from the merged distributions.  All samples that share the first instant go through the
kernels together.
"""
from __future__ import annotations

import random
from pathlib import Path
from typing import Optional, Sequence

import numpy as np
import torch

from contour_uncertainty.sampler.posterior_shape_model.psm import PosteriorShapeModelSampler, TABLE_STRIDE, load_psm
from contour_uncertainty.sampler.posterior_shape_model.psm_skew import cov_to3
from contour_uncertainty.sampler.posterior_shape_model.utils import index_to_flat
from cu_hip import ops

REC_M, REC_COVC, REC_G = 0, 96, 96 + 48 * 4      # record layout of cu_psm_setup (include/contour_hip.h)


class SequenceModel:
    """The two-instant PSM file + the per-first-instant level rows shared by both sequence samplers."""

    def __init__(self, sequence_psm_path: Path, nb_points: int):
        data = load_psm(sequence_psm_path)
        f = lambda k: torch.tensor(np.asarray(data[k]), dtype=torch.float)
        self.seq_mu, self.seq_Q = f("mu"), f("Q")
        self.seq_mean, self.seq_scale = f("scaler_mean"), f("scaler_scale")
        self.seq_X_train, self.seq_X_val = f("X_train"), f("X_val")
        k = self.k = nb_points
        x = self.seq_X_train.double()
        xbar = x.mean(0)
        self.cov0 = ((x - xbar).T @ (x - xbar) / x.shape[0]).float()
        self.xbar = xbar.float()
        self.tables, self.known_flat = [], []
        for first in (0, 1):
            g = index_to_flat(list(range(k)) if first == 0 else list(range(k, 2 * k)))
            t = list(range(k, 2 * k)) if first == 0 else list(range(k))
            row = np.zeros((1, TABLE_STRIDE), dtype=np.int32)
            row[0, 0], row[0, 1] = len(g), len(t)
            row[0, 2:2 + len(g)] = g
            row[0, 2 + 48:2 + 48 + len(t)] = t
            self.tables.append(torch.from_numpy(row))
            self.known_flat.append(g)
        self.rec_stride = ops.psm_record_floats([2 * k], [k])
        self._dev = {}

    def fixed_records(self):
        """Records of the FILE's (mu, Q) (sequence_sampler.py:83: no re-centring), f64 on the host, once."""
        k, p = self.k, 4 * self.k
        mu, Q = self.seq_mu.double().reshape(-1), self.seq_Q.double()
        recs = []
        for first in (0, 1):
            g = self.known_flat[first]
            mask = torch.zeros(p, dtype=torch.double)
            mask[g] = 1
            Qg = Q * mask[:, None]
            inv = torch.inverse(Qg.T @ Qg + torch.eye(p, dtype=torch.double))
            gain = Q @ inv @ Qg.T                                    # mu_c = mu + gain (s_g - mu_g)
            cov_c = (Q @ inv @ Q.T) * self.seq_scale.double()        # sigma2 = 1; `cov_c *= seq_scale` (:85)
            rec = torch.zeros(self.rec_stride, dtype=torch.double)
            rec[REC_M:REC_M + p] = mu
            t = list(range(k, 2 * k)) if first == 0 else list(range(k))
            for q, pt in enumerate(t):
                rec[REC_COVC + 4 * pt:REC_COVC + 4 * pt + 4] = cov_c[2 * pt:2 * pt + 2, 2 * pt:2 * pt + 2].reshape(-1)
                for a in range(2):
                    rec[REC_G + (2 * q + a) * len(g):REC_G + (2 * q + a + 1) * len(g)] = gain[2 * pt + a, g]
            recs.append(rec.float()[None])
        return recs

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = {
                "tables": [t.to(device) for t in self.tables], "mean": self.seq_mean.to(device),
                "scale": self.seq_scale.to(device), "cov0": self.cov0.to(device).contiguous(), "xbar": self.xbar.to(device),
                "fixed": [r.to(device) for r in self.fixed_records()]}
        return self._dev[key]

    def known(self, s_first: torch.Tensor, first: int) -> torch.Tensor:
        """(n, K, 2) contours of the first instant -> (n, 4K) flat two-instant vectors (other instant zero, :72-74)."""
        n, k = s_first.shape[0], self.k
        out = torch.zeros((n, 2, k, 2), dtype=torch.float32, device=s_first.device)
        out[:, first] = s_first
        return out.reshape(n, 4 * k)


class SequencePSMSampler(PosteriorShapeModelSampler):
    def __init__(self, psm_path: Path, sequence_psm_path: Path, levels: int = 3):
        super().__init__(psm_path, levels)
        self.seq = SequenceModel(sequence_psm_path, self.nb_points)
        for name in ("seq_mu", "seq_Q", "seq_mean", "seq_scale", "seq_X_train", "seq_X_val"):
            setattr(self, name, getattr(self.seq, name))

    def _second_instant(self, mu, cov, s_first, first):
        """-> mu_c (n,K,2), cov_c (K,2,2), mu_f (n,K,2), cov_f (K,2,2) of the OTHER instant (:83-94)."""
        dev = s_first.device
        d = self.seq.on(dev)
        k = self.nb_points
        mu_c, cov_c, mu_f, cov_f = ops.psm_condition(
            d["fixed"][first], d["tables"][first], k, self.seq.known(s_first, first), s_first.shape[0], d["mean"],
            d["scale"], mu.to(dev, torch.float32).reshape(1, 4 * k).contiguous(),
            cov_to3(cov.to(dev, torch.float32)).reshape(1, 2 * k, 3).contiguous())
        return mu_c, cov_c[0], mu_f, cov_f[0]

    def sample_pairs(self, mu: torch.Tensor, cov: torch.Tensor, firsts: torch.Tensor,
                     eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ALL ED/ES pairs in one launch set: mu (P,2,K,2), cov (P,2,K,2,2), firsts (P,n) in {0,1} = instant sampled
        first for sample i of pair p, eps (P,n,2,K,2) -> (P,n,2,K,2).

        Both orders are drawn for every sample (first = 0 and first = 1, each one batched launch over P x n) and the
        sample's own order is selected afterwards: twice the arithmetic of the per-pair loop, but no host
        synchronisation, no ragged groups and a launch count that does not grow with P."""
        p_, n, k = mu.shape[0], firsts.shape[1], self.nb_points
        dev = torch.device("cuda", torch.cuda.current_device()) if not mu.is_cuda else mu.device
        mu, cov = mu.to(dev, torch.float32), cov.to(dev, torch.float32)
        d = self.seq.on(dev)
        mu_p = mu.reshape(p_, 4 * k).contiguous()
        cov_p3 = cov_to3(cov).reshape(p_, 2 * k, 3).contiguous()
        eps = None if eps is None else eps.to(dev, torch.float32)
        branch = []
        for first in (0, 1):
            second = 1 - first
            s1 = self.sample_batch(mu[:, first], cov[:, first], n=n, eps=None if eps is None else eps[:, :, first])
            rec = d["fixed"][first].expand(p_, -1).contiguous()
            _, _, mu_f, cov_f = ops.psm_condition(rec, d["tables"][first], k, self.seq.known(s1.reshape(p_ * n, k, 2), first),
                                                  n, d["mean"], d["scale"], mu_p, cov_p3)
            e2 = None if eps is None else eps[:, :, second].reshape(p_ * n, 1, k, 2)
            s2 = self.sample_batch(mu_f, cov_f[:, None].expand(-1, n, -1, -1, -1).reshape(p_ * n, k, 2, 2), n=1, eps=e2)
            branch.append((s1, s2.reshape(p_, n, k, 2)))
        pick0 = (firsts.to(dev) == 0)[:, :, None, None]
        out = torch.empty((p_, n, 2, k, 2), dtype=torch.float32, device=dev)
        out[:, :, 0] = torch.where(pick0, branch[0][0], branch[1][1])
        out[:, :, 1] = torch.where(pick0, branch[0][1], branch[1][0])
        return out

    def sample_sequence(self, mu: torch.Tensor, cov: torch.Tensor, firsts: Sequence[int],
                        eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mu (2,K,2), cov (2,K,2,2), firsts[i] = instant sampled first for sample i, eps (n,2,K,2) -> (n,2,K,2)."""
        return self.sample_pairs(mu[None], cov[None], torch.as_tensor(list(firsts))[None],
                                 None if eps is None else eps[None])[0]

    def __call__(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor = None, n: int = 1,
                 debug_img=None) -> torch.Tensor:
        """mu (2,K,2), cov (2,K,2,2) -> (n,2,K,2); alpha is accepted and ignored like the reference (:48)."""
        firsts = [random.randint(0, 1) for _ in range(n)]
        return self.sample_sequence(mu, cov, firsts).to(mu.device)

    def sample_two_contours(self, mu, cov, alpha=None, first_sample=None, first_instant=0, debug_img=None):
        assert first_instant in [0, 1]
        second = 1 - first_instant
        k = self.nb_points
        dev = torch.device("cuda", torch.cuda.current_device()) if not mu.is_cuda else mu.device
        one = lambda m, c, a: (self.sample_batch(m, c, n=1) if a is None else
                               self.sample_batch_skew(m, c, a.to(dev)[None], n=1, skew_bits=0))     # psm.py:233-238
        if first_sample is None:
            s1 = one(mu[first_instant][None], cov[first_instant][None], None if alpha is None else alpha[first_instant])[0]
        else:
            s1 = first_sample.to(dev, torch.float32).reshape(1, k, 2)
        mu_c2, cov_c2, mu_f2, cov_f2 = self._second_instant(mu, cov, s1, first_instant)
        s2 = one(mu_f2, cov_f2[None], None if alpha is None else alpha[second])[:, 0]
        s = torch.zeros((2, k, 2), dtype=torch.float32, device=dev)
        s[first_instant], s[second] = s1[0], s2[0]
        # the reference also reports the first instant's (unused) conditional / merged rows; only the rows that feed
        # the sampler are produced here
        return {"mu_c": mu_c2[0], "cov_c": cov_c2, "mu_f": mu_f2[0], "cov_f": cov_f2, "s": s.to(mu.device),
                "second_instant": second}

    def sample_contour(self, mu, cov, n, debug_img=None):
        return PosteriorShapeModelSampler.__call__(self, mu, cov, n=n).squeeze()

    def sequence_transform(self, s):
        return ((s.reshape(1, -1) - self.seq_mean.to(s.device)) / self.seq_scale.to(s.device)).reshape(s.shape)

    def sequence_inverse_transform(self, s):
        return ((s.reshape(1, -1) * self.seq_scale.to(s.device)) + self.seq_mean.to(s.device)).reshape(s.shape)
