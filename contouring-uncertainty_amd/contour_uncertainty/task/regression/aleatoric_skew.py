"""``SkewUncertaintyTask`` (reference task/regression/aleatoric_skew.py:25-127)."""
from __future__ import annotations

from pathlib import Path
from typing import Any, List, Tuple

import numpy as np
import torch

from contour_uncertainty._compat import ContourTags, Tags, to_absolute_path
from contour_uncertainty.data.config import BatchResult
from contour_uncertainty.task.regression.aleatoric import AleatoricUncertaintyTask
from contour_uncertainty.utils.posterior_stats import sample_moments_pooled, total_moments


class SkewUncertaintyTask(AleatoricUncertaintyTask):
    def __init__(self, psm_path: str = None, seq_psm_path=None, skew_indices: List[int] = None, *args, **kwargs):
        super().__init__(psm_path, seq_psm_path, *args, **kwargs)
        k = self.hparams.data_params.out_shape[0]
        self.skew_indices = list(range(k)) if skew_indices is None else list(skew_indices)

    def _build_sampler(self):
        from contour_uncertainty.sampler.posterior_shape_model.psm_skew import SkewPosteriorShapeModelSampler
        from contour_uncertainty.sampler.posterior_shape_model.psm_skew_sequence import SequenceSkewPSMSampler
        if self.hparams.sequence_sampler:
            # the reference drops skew_indices here (aleatoric_skew.py:37-38): every point is a skew point
            return SequenceSkewPSMSampler(sequence_psm_path=Path(to_absolute_path(self.hparams.seq_psm_path)),
                                          psm_path=Path(to_absolute_path(self.hparams.psm_path)))
        # the reference passes the raw (possibly None) skew_indices (aleatoric_skew.py:41-42);
        # the resolved list is what its sampler needs
        return SkewPosteriorShapeModelSampler(psm_path=Path(to_absolute_path(self.hparams.psm_path)),
                                              skew_indices=self.skew_indices)

    def predict(self, img) -> Tuple:  # noqa: D102
        raise NotImplementedError

    def sample(self, mu, cov, alpha, T):
        """(N, T_e, K, .) -> (N, T_e, T, K, 2)  (reference aleatoric_skew.py:48-53)"""
        from contour_uncertainty.sampler.posterior_shape_model.psm_skew_sequence import SequenceSkewPSMSampler
        if isinstance(self.sampler, SequenceSkewPSMSampler):      # an ED/ES pair is one unit: (2, T, K, 2) per t_e
            import random
            firsts = torch.tensor([[random.randint(0, 1) for _ in range(T)] for _ in range(mu.shape[1])])
            out = self.sampler.sample_pairs(mu.transpose(0, 1), cov.transpose(0, 1), alpha.transpose(0, 1), firsts)
            return out.permute(2, 0, 1, 3, 4).cpu().numpy()                      # (T_e, T, 2, K, 2) -> (2, T_e, T, K, 2)
        # frames are independent: all (frame, t_e) pairs and all T samples in one launch
        n, te, k = mu.shape[:3]
        out = self.sampler.sample_batch(mu.reshape(-1, k, 2), cov.reshape(-1, k, 2, 2), alpha.reshape(-1, k, 2), n=T)
        return out.reshape(n, te, T, k, 2).cpu().numpy()

    def _predict_step(self, batch: Any) -> BatchResult:
        """reference aleatoric_skew.py:55-127 (mode / u-map projection is 8(f) rank 3: taken from the datamodule fn)."""
        img = batch[Tags.img]
        contour = batch[ContourTags.contour]
        gt = batch[Tags.gt].cpu().numpy() if Tags.gt in batch.keys() else None
        n = img.shape[0]
        mu, cov, alpha = self.predict(img)
        contour_samples = self.sample(mu, cov, alpha, 25)            # T hard-coded in the reference (:63)
        mu_np, cov_al, cov_ep = total_moments(mu, cov)
        cov_np = cov_ep + cov_al
        alpha_np = alpha.mean(dim=1).cpu().numpy()
        post_mu, post_cov = sample_moments_pooled(contour_samples)
        mode, umap = mu_np, None
        if self.skew_umap_fn is not None:
            mm, uu = zip(*[self.skew_umap_fn(mu_np[i], cov_np[i], alpha_np[i], self.hparams.data_params.labels)
                           for i in range(n)])
            mode, umap = np.array(mm), np.array(uu)
        pred, pred_samples = self.convert_to_mask(mode, img.shape, contour_samples)
        return BatchResult(id=batch.get(Tags.id), labels=self.hparams.data_params.labels, img=img,
                           contour=contour.cpu().numpy(), gt=gt, mu=mu_np, mode=mode, cov=cov_np,
                           contour_samples=contour_samples, pred_samples=pred_samples, pred=pred,
                           uncertainty_map=umap, alpha=alpha_np, post_mu=post_mu, post_cov=post_cov)
