"""reference contour_uncertainty/results/clinical/utils.py:3-20"""
import numpy as np


def aleatoric_epistemic_uncertainty(metric_mc):
    """metric_mc (T_e, T_a): Monte-Carlo values of a clinical metric (NaN = rejected sample) ->
    (mean, aleatoric std = mean over T_e of the std over T_a, epistemic std = std over T_e of the means, their sum)"""
    metric_mc = np.asarray(metric_mc, dtype=float)
    assert metric_mc.ndim == 2, "metric_mc should have 2 dimensions, current shape is {}".format(metric_mc.shape)
    metric_means = np.nanmean(metric_mc, axis=-1)
    metric_vars = np.nanstd(metric_mc, axis=-1)
    metric_mean = np.nanmean(metric_means)
    epistemic_var = np.nanstd(metric_means)
    aleatoric_var = np.nanmean(metric_vars)
    metric_variance = epistemic_var + aleatoric_var
    return metric_mean, aleatoric_var, epistemic_var, metric_variance
