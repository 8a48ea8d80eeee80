"""``SequencePSMSampler`` (reference sampler/posterior_shape_model/sequence_sampler.py:13-160): ED/ES two-instant
conditioning on an 84-dimensional PSM.  Not built yet; ``task.sequence_sampler`` defaults to
False in every dsnt config."""


class SequencePSMSampler:
    def __init__(self, sequence_psm_path=None, psm_path=None):
        raise NotImplementedError("SequencePSMSampler is not part of this round (task.sequence_sampler=False is the "
                                  "default of config/task/dsnt-*.yaml)")
