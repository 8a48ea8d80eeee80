"""``AleatoricUncertaintyTask`` (reference task/regression/aleatoric.py:24-144): sampler construction, ``sample``,
``_predict_step`` (aleatoric / epistemic covariance split, posterior statistics of the samples), ``get_cov_matrix``."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Tuple

import numpy as np
import torch

from contour_uncertainty._compat import ContourTags, Tags, to_absolute_path
from contour_uncertainty.data.config import BatchResult
from contour_uncertainty.task.regression.contour_uncertainty import ContourUncertaintyTask
from contour_uncertainty.utils.posterior_stats import sample_moments_per_member, total_moments


class AleatoricUncertaintyTask(ContourUncertaintyTask):
    def __init__(self, psm_path: str = None, seq_psm_path=None, sequence_sampler=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        self._sampler = None

    # the PSM files are only needed at predict time: build the sampler on first use
    def _build_sampler(self):
        from contour_uncertainty.sampler.posterior_shape_model.psm import PosteriorShapeModelSampler
        from contour_uncertainty.sampler.posterior_shape_model.sequence_sampler import SequencePSMSampler
        if self.hparams.sequence_sampler:
            return SequencePSMSampler(sequence_psm_path=Path(to_absolute_path(self.hparams.seq_psm_path)),
                                      psm_path=Path(to_absolute_path(self.hparams.psm_path)))
        return PosteriorShapeModelSampler(psm_path=Path(to_absolute_path(self.hparams.psm_path)))

    @property
    def sampler(self):
        if self._sampler is None:
            self._sampler = self._build_sampler()
        return self._sampler

    @sampler.setter
    def sampler(self, s):
        self._sampler = s

    def predict(self, img) -> Tuple:  # noqa: D102
        raise NotImplementedError

    def sample(self, mu: torch.Tensor, cov: torch.Tensor, t_a: int):
        """mu (N, T_e, K, 2), cov (N, T_e, K, 2, 2) -> contour samples (N, T_e, T_a, K, 2)  (reference :54-78)."""
        from contour_uncertainty.sampler.posterior_shape_model.sequence_sampler import SequencePSMSampler
        n = mu.shape[0]
        if isinstance(self.sampler, SequencePSMSampler):
            # an ED/ES pair per t_e member: all pairs in one launch set; the first instant of each sample is drawn with
            # ``random.randint`` like the reference (sequence_sampler.py:47)
            import random
            firsts = torch.tensor([[random.randint(0, 1) for _ in range(t_a)] for _ in range(mu.shape[1])])
            out = self.sampler.sample_pairs(mu.transpose(0, 1), cov.transpose(0, 1), firsts)       # (T_e, T_a, 2, K, 2)
            return out.permute(2, 0, 1, 3, 4).cpu().numpy()
        # frames are independent: the batched GPU sampler takes all (frame, t_e) pairs in one call
        out = self.sampler.sample_batch(mu.reshape(-1, *mu.shape[2:]), cov.reshape(-1, *cov.shape[2:]), n=t_a)
        return out.reshape(n, mu.shape[1], t_a, mu.shape[2], 2).cpu().numpy()

    def _predict_step(self, batch: Any) -> BatchResult:
        """reference aleatoric.py:80-135"""
        img = batch[Tags.img]
        contour = batch[ContourTags.contour]
        gt = batch[Tags.gt].cpu().numpy() if Tags.gt in batch.keys() else None
        n = img.shape[0]
        mu, cov = self.predict(img)                      # (N, T_e, K, 2), (N, T_e, K, 2, 2) on CPU
        contour_samples = self.sample(mu, cov, self.hparams.t_a)
        mu_np, cov_al, cov_ep = total_moments(mu, cov)
        cov_np = cov_al + cov_ep
        post_mu, post_cov = sample_moments_per_member(contour_samples)
        pred, pred_samples = self.convert_to_mask(mu_np, img.shape, contour_samples)
        pred = pred_samples.mean(axis=(1, 2)).squeeze().round().astype(int)
        umap = None
        if self.umap_fn is not None:
            umap = np.array([self.umap_fn(mu_np[i], cov_np[i], self.hparams.data_params.labels) for i in range(n)])
        return BatchResult(id=batch.get(Tags.id), labels=self.hparams.data_params.labels, img=img,
                           contour=contour.cpu().numpy(), gt=gt, mu=mu_np, mode=mu_np, cov=cov_np,
                           contour_samples=contour_samples, pred_samples=pred_samples, pred=pred,
                           uncertainty_map=umap, post_mu=post_mu, post_cov=post_cov)

    def get_cov_matrix(self, var_x, var_y, covar_xy=0):
        """reference aleatoric.py:138-144"""
        Sigma = torch.zeros((var_x.shape[0], var_x.shape[1], 2, 2), device=var_x.device)
        Sigma[:, :, 0, 0] = var_x
        Sigma[:, :, 0, 1] = covar_xy
        Sigma[:, :, 1, 0] = covar_xy
        Sigma[:, :, 1, 1] = var_y
        return Sigma
