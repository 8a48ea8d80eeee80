"""``SkewPosteriorShapeModelSampler`` (reference sampler/posterior_shape_model/psm_skew.py:162-503): grid-product
categorical sampling with skew-normal point distributions.  Not built yet.  Note that the
reference implementation itself crashes for partially-skewed configs (``merge_gaussian_priors`` is undefined,
psm_skew.py:329)."""


class SkewPosteriorShapeModelSampler:
    def __init__(self, psm_path=None, skew_indices=None, levels: int = 3):
        raise NotImplementedError("SkewPosteriorShapeModelSampler is not part of this round; the Gaussian "
                                  "PosteriorShapeModelSampler is (cu_psm_sample_gauss)")
