"""``SequenceSkewPSMSampler`` on the MI355X kernels (reference sampler/posterior_shape_model/psm_skew_sequence.py:21-166).

Per ED/ES pair: the 84-dimensional PSM is re-centred on the predicted pair (``cu_psm_setup``, P = 84, :66); the first
instant is drawn with the skew grid sampler; the conditional N(mu_c, cov_c) of the other instant given that draw
(``cu_psm_condition``) multiplies the other instant's skew tables (:86-99), and the other instant is drawn from those
products, anchors included (``use_initial_pdf``).  Quirk kept: the second instant's tables use alpha as given (:90),
the first instant's use alpha_y negated (psm_skew.py:232).
"""
from __future__ import annotations

import random
from pathlib import Path
from typing import List, Optional, Sequence

import torch

from contour_uncertainty.sampler.posterior_shape_model.psm_skew import SkewPosteriorShapeModelSampler
from contour_uncertainty.sampler.posterior_shape_model.sequence_sampler import SequenceModel
from cu_hip import ops


class SequenceSkewPSMSampler(SkewPosteriorShapeModelSampler):
    def __init__(self, psm_path: Path, sequence_psm_path: Path, levels: int = 3, skew_indices: List[int] = None):
        super().__init__(psm_path, levels, skew_indices)
        self.seq = SequenceModel(sequence_psm_path, self.nb_points)
        for name in ("seq_mu", "seq_Q", "seq_mean", "seq_scale", "seq_X_train", "seq_X_val"):
            setattr(self, name, getattr(self.seq, name))

    def sample_pairs(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor, firsts: torch.Tensor,
                     eps: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ALL ED/ES pairs in one launch set: mu (P,2,K,2), cov (P,2,K,2,2), alpha (P,2,K,2), firsts (P,n) in {0,1},
        eps (P,n,2,K,3), u (P,n,2,K) -> (P,n,2,K,2).  Both orders are drawn for every sample and the sample's own order
        is selected afterwards (see ``SequencePSMSampler.sample_pairs``)."""
        p_, n, k = mu.shape[0], firsts.shape[1], self.nb_points
        dev = torch.device("cuda", torch.cuda.current_device()) if not mu.is_cuda else mu.device
        mu, cov, alpha = (t.to(dev, torch.float32) for t in (mu, cov, alpha))
        d = self.seq.on(dev)
        mu_flat = mu.reshape(p_, 4 * k).contiguous()
        pick = lambda t, inst: None if t is None else t.to(dev, torch.float32)[:, :, inst]
        branch = []
        for first in (0, 1):
            second = 1 - first
            s1 = self.sample_batch(mu[:, first], cov[:, first], alpha[:, first], n=n, eps=pick(eps, first),
                                   u=pick(u, first))
            rec = ops.psm_setup(mu_flat, d["cov0"], d["xbar"], d["mean"], d["scale"], d["tables"][first], [1.0],
                                self.seq.rec_stride)
            mu_c, cov_c, _, _ = ops.psm_condition(rec, d["tables"][first], k, self.seq.known(s1.reshape(p_ * n, k, 2), first),
                                                  n, d["mean"], d["scale"])
            s2 = self.sample_batch(mu[:, second], cov[:, second], alpha[:, second], n=n, eps=pick(eps, second),
                                   u=pick(u, second), prior_mu=mu_c.reshape(p_, n, k, 2),
                                   prior_cov=cov_c[:, None].expand(-1, n, -1, -1, -1), use_initial_pdf=True,
                                   flip_alpha_y=False)
            branch.append((s1, s2))
        pick0 = (firsts.to(dev) == 0)[:, :, None, None]
        out = torch.empty((p_, n, 2, k, 2), dtype=torch.float32, device=dev)
        out[:, :, 0] = torch.where(pick0, branch[0][0], branch[1][1])
        out[:, :, 1] = torch.where(pick0, branch[0][1], branch[1][0])
        return out

    def sample_sequence(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor, firsts: Sequence[int],
                        eps: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mu (2,K,2), cov (2,K,2,2), alpha (2,K,2), eps (n,2,K,3), u (n,2,K) -> (n,2,K,2)."""
        return self.sample_pairs(mu[None], cov[None], alpha[None], torch.as_tensor(list(firsts))[None],
                                 None if eps is None else eps[None], None if u is None else u[None])[0]

    def __call__(self, mu: torch.Tensor, cov: torch.Tensor, alpha: torch.Tensor, n: int = 1, debug_img=None,
                 progress_bar=False) -> torch.Tensor:
        """-> (2, n, K, 2)  (psm_skew_sequence.py:34-48)."""
        firsts = [random.randint(0, 1) for _ in range(n)]
        return self.sample_sequence(mu, cov, alpha, firsts).permute(1, 0, 2, 3).to(mu.device)

    def sample_two_contours(self, mu, cov, alpha, first_sample=None, first_instant=0, debug_img=None):
        return self.sample_sequence(mu, cov, alpha, [first_instant])[0].to(mu.device)

    def sequence_transform(self, s):
        return ((s.reshape(1, -1) - self.seq_mean.to(s.device)) / self.seq_scale.to(s.device)).reshape(s.shape)

    def sequence_inverse_transform(self, s):
        return ((s.reshape(1, -1) * self.seq_scale.to(s.device)) + self.seq_mean.to(s.device)).reshape(s.shape)
