"""The skew sampler's three window modes must draw the SAME cells (tuning build: CU_PSM_MERGED_WINDOW = 0 conditional Gaussian's
window, 1 + window of the product, 2 + narrow box first with the certainty check), on the c5 bench inputs.

    CONTOUR_HIP_LIB=$PWD/contouring-uncertainty_amd/libcontour_hip_tuning.so python tools/psm_window_check.py
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import numpy as np
import torch
from contour_uncertainty.sampler.posterior_shape_model.psm_skew import SkewPosteriorShapeModelSampler

dev = torch.device("cuda", 0)
psm_path = ROOT / "tests" / "golden" / "camus-cont_psm_11_no_std.npz"
psm = dict(np.load(psm_path))
F, NS = 64, 1024
g = torch.Generator().manual_seed(100)
idx = torch.randint(0, psm["X_val"].shape[0], (F,), generator=g)
mu = torch.stack([torch.tensor(psm["X_val"][i] + psm["scaler_mean"]).float().reshape(21, 2) for i in idx.tolist()])
a = torch.randn(F, 21, 2, 2, generator=g)
out = {}
for scale in (6.0, 0.5, 40.0):
    cov = a @ a.transpose(-1, -2) * scale + torch.eye(2) * 2.0
    alpha = torch.randn(F, 21, 2, generator=g) * 2.0
    sk = SkewPosteriorShapeModelSampler(psm_path)
    res = []
    for mode in ("0", "1", "2"):
        os.environ["CU_PSM_MERGED_WINDOW"] = mode
        res.append(sk.sample_batch(mu.to(dev), cov.to(dev), alpha.to(dev), n=NS, seed=7).cpu())
    d01 = int((res[0] != res[1]).any(-1).sum()), int((res[1] != res[2]).any(-1).sum())
    print(f"prediction covariance scale {scale}: {F * NS * 21} points; differing points mode 0 vs 1: {d01[0]}, mode 1 vs 2: {d01[1]}; "
          f"max |diff| {float((res[0] - res[1]).abs().max()):.3g} / {float((res[1] - res[2]).abs().max()):.3g} px")
