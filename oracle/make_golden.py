"""Generate tests/golden/*.npz from the REFERENCE's own importable leaf modules.

Run only in the build container (the reference never travels):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

It imports ``contour_uncertainty.*`` from /root/reference, feeds it seeded inputs and stores inputs + outputs
(+ gradients) as small fixtures.  Weights for the network fixtures come from ``oracle.unet.init_unet_state`` (seeded
``torch.Generator``), are loaded into the reference ``UNet`` with ``strict=True`` (which also pins parameter names and
shapes) and are NOT stored: tests regenerate them from the same seed.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(REF))
os.environ.setdefault("MPLBACKEND", "Agg")

from oracle import unet as OU  # noqa: E402

OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def stats(t: torch.Tensor):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


# ----------------------------------------------------------------------------------------------- (i) dsnt head
def gen_dsnt():
    from contour_uncertainty.task.regression.dsnt import utils as R
    out = {}
    for size, n, k, scale, seed in ((16, 2, 3, 3.0, 1), (64, 2, 5, 6.0, 2), (256, 1, 2, 8.0, 3)):
        g = torch.Generator().manual_seed(seed)
        logits = (torch.randn(n, k, size, size, generator=g) * scale).requires_grad_(True)
        hm = R.flat_softmax(logits)
        coords, var, covar = R.dsnt(hm)
        px = R.normalized_to_pixel_coordinates(coords, size)
        # upstream grads to exercise backward
        gc = torch.randn(coords.shape, generator=g)
        gv = torch.randn(var.shape, generator=g)
        gcv = torch.randn(covar.shape, generator=g)
        (coords * gc).sum().add((var * gv).sum()).add((covar * gcv).sum()).backward()
        tag = f"s{size}"
        out[f"{tag}_logits"] = npy(logits)
        out[f"{tag}_coords"] = npy(coords)
        out[f"{tag}_var"] = npy(var)
        out[f"{tag}_covar"] = npy(covar)
        out[f"{tag}_pixel"] = npy(px)
        out[f"{tag}_g_coords"] = npy(gc)
        out[f"{tag}_g_var"] = npy(gv)
        out[f"{tag}_g_covar"] = npy(gcv)
        if size <= 64:
            out[f"{tag}_dlogits"] = npy(logits.grad)
        else:
            out[f"{tag}_dlogits_stats"] = stats(logits.grad)
            out[f"{tag}_dlogits_row"] = npy(logits.grad[0, 0, 100])
    out["linspace4"] = npy(R.normalized_linspace(4))          # docstring KAT utils.py:54-58
    np.savez_compressed(OUT / "dsnt_head.npz", **out)


# ----------------------------------------------------------------------------------------------- (ii) NLL heads
def rand_spd(m, g, lo=4.0, hi=400.0):
    a = torch.randn(m, 2, 2, generator=g)
    s = a @ a.transpose(-1, -2)
    s = s / s.diagonal(dim1=-2, dim2=-1).mean(-1)[:, None, None]
    return s * (lo + (hi - lo) * torch.rand(m, 1, 1, generator=g)) + 0.05 * torch.eye(2)


def gen_nll():
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal as BSN
    g = torch.Generator().manual_seed(7)
    m = 48
    mu = (torch.rand(m, 2, 1, generator=g) * 200 + 20)
    y = mu + torch.randn(m, 2, 1, generator=g) * 6
    cov = rand_spd(m, g)
    alpha = torch.randn(m, 2, 1, generator=g) * 3
    # edge rows: near-singular, isotropic (repeated eigenvalue), huge |alpha| with Phi -> 0, alpha = 0
    cov[0] = torch.tensor([[25.0, 24.99], [24.99, 25.0]])
    cov[1] = torch.tensor([[30.0, 1e-3], [1e-3, 30.0001]])
    alpha[2] = torch.tensor([[40.0], [-35.0]])
    y[2] = mu[2] - torch.tensor([[9.0], [-7.0]])
    alpha[3] = 0.0
    out = {"mu": npy(mu), "y": npy(y), "cov": npy(cov), "alpha": npy(alpha)}

    # skew NLL (bivariateskewnormal.py:51-61) + grads
    mu_r, cov_r, al_r = (t.clone().requires_grad_(True) for t in (mu, cov, alpha))
    nll, t1, t2, t3 = BSN.nll(y, mu_r, cov_r, al_r)
    nll.mean().backward()
    out.update(skew_nll=npy(nll), skew_t1=npy(t1), skew_t2=npy(t2), skew_t3=npy(t3),
               skew_dmu=npy(mu_r.grad), skew_dcov=npy(cov_r.grad), skew_dalpha=npy(al_r.grad))

    # Gaussian NLL written exactly like dsnt_al.py:64-71 (literal broadcast) + grads
    mu_r, cov_r = (t.clone().requires_grad_(True) for t in (mu, cov))
    w_log, w_mse = 1.0, 1.0
    lt1 = w_log * torch.log(torch.det(cov_r))
    lt2 = w_mse * (((mu_r - y).transpose(-1, -2) @ torch.inverse(cov_r)) @ (mu_r - y))
    loss = (lt1 + lt2).mean()
    loss.backward()
    out.update(gauss_loss=npy(loss), gauss_t1_mean=npy(lt1.mean()), gauss_t2_mean=npy(lt2.mean()),
               gauss_dmu=npy(mu_r.grad), gauss_dcov=npy(cov_r.grad))

    # densities on a small grid (bivariatenormal.py:15-36, bivariateskewnormal.py:19-34)
    from contour_uncertainty.distributions.bivariatenormal import BivariateNormal as BN
    xx, yy = np.meshgrid(np.linspace(0, 64, 64), np.linspace(0, 64, 64))
    pos = torch.tensor(np.dstack((xx, yy))).float()
    loc = torch.tensor([30.0, 25.0])
    c = torch.tensor([[25.0, 4.0], [4.0, 50.0]])
    al = torch.tensor([3.0, -1.5])
    out.update(grid_pos=npy(pos), grid_loc=npy(loc), grid_cov=npy(c), grid_alpha=npy(al),
               grid_gauss_logpdf=npy(BN.logpdf(pos, loc, c)), grid_skew_logpdf=npy(BSN.logpdf(pos, loc, c, al)))
    # reference's own manual KAT constants (bivariatenormal.py:98-103): pdf vs scipy
    from scipy.stats import multivariate_normal
    mu_k = torch.tensor([100.0, 100.0])
    cov_k = torch.tensor([[25.0, 4.0], [4.0, 50.0]])
    pts = torch.tensor([[100.0, 100.0], [104.0, 97.0], [90.0, 111.0]])
    out.update(kat_pts=npy(pts), kat_ref=npy(BN.pdf(pts, mu_k, cov_k)),
               kat_scipy=multivariate_normal(mean=mu_k.numpy(), cov=cov_k.numpy()).pdf(pts.numpy()))
    np.savez_compressed(OUT / "nll_heads.npz", **out)


# ----------------------------------------------------------------------------------------------- (iii) network
def ref_unet(spec: OU.UNetSpec, sd, bottleneck_out):
    from contour_uncertainty.models.nnUnet.unet2 import UNet
    n = len(spec.strides)
    net = UNet((spec.in_channels, 0, 0), (spec.num_classes, 0, 0), [256, 256], [[3, 3]] * n,
               [[s, s] for s in spec.strides], bottleneck_out=bottleneck_out)
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    # state_dict order must match too
    assert list(net.state_dict().keys()) == list(sd.keys()), "parameter order differs from the reference"
    return net


def gen_unet_small():
    """4-stage net (32,64,128,256) at 32x32, N=2, K=5: full logits + bottleneck + per-parameter grad stats."""
    spec = OU.UNetSpec(in_channels=1, num_classes=5, strides=(1, 2, 2, 2))
    g = torch.Generator().manual_seed(11)
    sd = OU.init_unet_state(spec, g)
    # make norm affine / biases non-trivial so their gradients and use are exercised
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    x = torch.rand(2, 1, 32, 32, generator=g)
    net = ref_unet(spec, sd, bottleneck_out=True)
    net.train()
    logits, bott = net(x)
    gl = torch.randn(logits.shape, generator=g)
    gb = torch.randn(bott.shape, generator=g)
    ((logits * gl).sum() + (bott * gb).sum()).backward()
    out = {"x": npy(x), "logits": npy(logits), "bottleneck": npy(bott), "g_logits": npy(gl), "g_bott": npy(gb)}
    names, gstats, ghead = [], [], []
    for name, p in net.named_parameters():
        if p.grad is None:
            continue
        names.append(name)
        gstats.append(stats(p.grad))
        ghead.append(npy(p.grad.flatten()[:8]))
    out["grad_names"] = np.array(names)
    out["grad_stats"] = np.stack(gstats)
    out["grad_head"] = np.stack([np.pad(h, (0, 8 - len(h))) for h in ghead])
    out["no_grad_names"] = np.array([n for n, p in net.named_parameters() if p.grad is None])
    np.savez_compressed(OUT / "unet_small.npz", **out)


def gen_unet_full():
    """The real 8-stage config (unet2.yaml) at 256x256, N=1: head outputs, logits statistics, slices."""
    from contour_uncertainty.task.regression.dsnt import utils as R
    from contour_uncertainty.models.nnUnet.unet2 import ConfidenceNet
    spec = OU.UNetSpec()
    g = torch.Generator().manual_seed(0)
    sd = OU.init_unet_state(spec, g)
    ssd = OU.init_confidence_state(42, g)
    x = torch.rand(1, 1, 256, 256, generator=g)
    net = ref_unet(spec, sd, bottleneck_out=True)
    head = ConfidenceNet(42)
    head.load_state_dict(ssd, strict=True)
    with torch.no_grad():
        logits, bott = net(x)
        a = head(bott)
        hm = R.flat_softmax(logits)
        coords, var, covar = R.dsnt(hm)
    out = {"x_seed": np.array(0), "logits_stats": stats(logits), "logits_row": npy(logits[0, :, 128, :]),
           "logits_col": npy(logits[0, :, :, 77]), "bottleneck": npy(bott), "alpha_raw": npy(a),
           "coords": npy(coords), "var": npy(var), "covar": npy(covar),
           "n_params_unet": np.array(sum(v.numel() for v in sd.values())),
           "n_params_skew": np.array(sum(v.numel() for v in ssd.values())),
           "param_names": np.array(list(sd.keys()))}
    np.savez_compressed(OUT / "unet_full.npz", **out)


def gen_step():
    """Two dsnt-skew and two dsnt-al training steps (6-stage net at 64x64, N=2 = BASELINE config c1 shape) composed
    from the reference's leaf modules exactly like dsnt_skew.py:61-104 / dsnt_al.py:45-74, with torch.optim.Adam."""
    from contour_uncertainty.task.regression.dsnt import utils as R
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal as BSN
    from contour_uncertainty.models.nnUnet.unet2 import ConfidenceNet
    from oracle.step import synthetic_batch
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    out = {}
    for task in ("dsnt-skew", "dsnt-al"):
        g = torch.Generator().manual_seed(0)
        sd = OU.init_unet_state(spec, g)
        net = ref_unet(spec, sd, bottleneck_out=(task == "dsnt-skew"))
        mods = [net]
        if task == "dsnt-skew":
            ssd = OU.init_confidence_state(42, g)
            head = ConfidenceNet(42)
            head.load_state_dict(ssd, strict=True)
            mods.append(head)
        params = [p for m in mods for p in m.parameters()]
        opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-3)
        img, contour = synthetic_batch(2, 64, 21, seed=1234)
        logs_all = []
        for it in range(2):
            opt.zero_grad()
            size = img.shape[2]
            if task == "dsnt-skew":
                heat, feats = net(img)
                alpha = head(feats).view(2, 21, 2)
            else:
                heat = net(img)
            hm = R.flat_softmax(heat)
            coords, var, covar = R.dsnt(hm)
            mu = R.normalized_to_pixel_coordinates(coords, size)
            pvar = var * (size / 2) ** 2
            pcov = covar * (size / 2) ** 2
            S = torch.zeros(2, 21, 2, 2)
            S[:, :, 0, 0] = pvar[..., 0]
            S[:, :, 0, 1] = pcov
            S[:, :, 1, 0] = pcov
            S[:, :, 1, 1] = pvar[..., 1]
            dist = R.euclidean_losses(mu, contour).mean()
            mu_f = mu.flatten(0, 1).unsqueeze(-1)
            y_f = contour.flatten(0, 1).unsqueeze(-1)
            S_f = S.flatten(0, 1)
            if task == "dsnt-skew":
                a_f = alpha.flatten(0, 1).unsqueeze(-1)
                nll, t1, t2, t3 = BSN.nll(y_f, mu_f, S_f, a_f)
                loss = nll.mean()
                logs = [loss, dist, t1.mean(), t2.mean(), t3.mean(), torch.norm(a_f, dim=-1).mean()]
            else:
                t1 = torch.log(torch.det(S_f))
                t2 = ((mu_f - y_f).transpose(-1, -2) @ torch.inverse(S_f)) @ (mu_f - y_f)
                loss = (t1 + t2).mean()
                logs = [loss, dist, t1.mean(), t2.mean()]
            loss.backward()
            if it == 0:
                out[f"{task}_mu0"] = npy(mu)
                out[f"{task}_sigma0"] = npy(S)
                if task == "dsnt-skew":
                    # raw head output of the step (dsnt_skew.py:68-71, skew_indices = all 21) and what predict_on_batch
                    # returns for the same weights: alpha_y negated (dsnt_skew.py:164)
                    out["dsnt-skew_alpha0"] = npy(alpha)
                    a_pred = alpha.detach().clone()
                    a_pred[..., 1] = -a_pred[..., 1]
                    out["dsnt-skew_alpha0_predict"] = npy(a_pred)
                gn, gs = [], []
                for m, pre in zip(mods, ("model.", "skew_block.")):
                    for name, p in m.named_parameters():
                        if p.grad is not None:
                            gn.append(pre + name)
                            gs.append(stats(p.grad))
                out[f"{task}_grad_names"] = np.array(gn)
                out[f"{task}_grad_stats"] = np.stack(gs)
            opt.step()
            logs_all.append([float(v) for v in logs])
        out[f"{task}_logs"] = np.array(logs_all)
        # a few post-update weights
        out[f"{task}_w_out"] = npy(net.output_block.conv.weight)
        out[f"{task}_w_in"] = npy(net.input_block.conv1.conv.weight)
        out[f"{task}_b_bott"] = npy(net.bottleneck.conv2.conv.bias)
    np.savez_compressed(OUT / "train_step.npz", **out)


# ----------------------------------------------------------------------------------------------- (iv) PSM math
def gen_psm():
    from contour_uncertainty.sampler.posterior_shape_model.posteriorshapemodel import pca, posterior_shape_model
    from contour_uncertainty.sampler.posterior_shape_model.utils import index_to_flat
    from contour_uncertainty.sampler.sampler import Sampler
    psm = np.load(REF / "camus-cont_psm_11_no_std.npy", allow_pickle=True).item()
    seq = np.load(REF / "camus-cont_sequence_psm_11_no_std.npy", allow_pickle=True).item()
    # plain arrays of the shipped PSM files (data, not code) so the product can load them without pickle
    np.savez_compressed(OUT / "camus-cont_psm_11_no_std.npz", **{k: np.asarray(v) for k, v in psm.items()})
    np.savez_compressed(OUT / "camus-cont_sequence_psm_11_no_std.npz", **{k: np.asarray(v) for k, v in seq.items()})

    X = torch.tensor(np.asarray(psm["X_train"])).float()
    sm = np.asarray(psm["scaler_mean"])
    mu_pred = torch.tensor((np.asarray(psm["X_val"])[3] + 0.0)).float().reshape(-1, 1)     # a val shape as "prediction"
    mu_p, Q = pca(X, mu_pred)
    out = {"pca_mu": npy(mu_p), "pca_QQt": npy(Q @ Q.T), "pca_Qabs_colnorm": npy(Q.norm(dim=0)),
           "scaler_mean": sm, "mu_pred": npy(mu_pred)}
    init, order = Sampler.get_points_order(21, levels=3)
    out["order_init"] = np.array(init)
    for i, lv in enumerate(order):
        out[f"order_l{i}"] = np.array(lv)
    g = torch.Generator().manual_seed(5)
    s = mu_pred + torch.randn(42, 1, generator=g) * 2.0
    known = list(init)
    for i, lv in enumerate(order):
        idx = index_to_flat(sorted(known))
        mu_c, cov_c = posterior_shape_model(s, idx, mu_pred, Q, sigma2=1)
        out[f"psm_l{i}_idx"] = np.array(idx)
        out[f"psm_l{i}_mu"] = npy(mu_c)
        out[f"psm_l{i}_cov"] = npy(cov_c)
        known += lv
    idx = index_to_flat(sorted(known))
    mu_c, cov_c = posterior_shape_model(s, idx, mu_pred, Q, sigma2=0.001)
    out["psm_final_idx"] = np.array(idx)
    out["psm_final_mu"] = npy(mu_c)
    out["psm_final_cov"] = npy(cov_c)
    out["psm_s"] = npy(s)
    np.savez_compressed(OUT / "psm_math.npz", **out)


# ----------------------------------------------------------------------------------------------- (v) skew grid sampler
def gen_skew_grid():
    """Pieces of SkewPosteriorShapeModelSampler / SequenceSkewPSMSampler that ARE importable: the density tables of
    `numerical_sampling` (psm_skew.py:60-88), rvs_fast with the generator state pinned, and the two-instant PSM
    conditional (sequence_sampler.py:83-86)."""
    from torch.distributions import MultivariateNormal
    from contour_uncertainty.distributions.bivariatenormal import BivariateNormal
    from contour_uncertainty.distributions.bivariateskewnormal import BivariateSkewNormal
    from contour_uncertainty.sampler.posterior_shape_model.posteriorshapemodel import posterior_shape_model, pca
    from contour_uncertainty.sampler.posterior_shape_model.utils import index_to_flat
    x = torch.linspace(0, 255, 256)
    X, Y = torch.meshgrid(x, x, indexing="ij")
    grid = torch.stack([X, Y], dim=-1)
    cases = [   # mu1, cov1, alpha1, mu2, cov2
        ([120.3, 88.6], [[30.0, 8.0], [8.0, 18.0]], [2.5, -1.0], [123.0, 91.5], [[9.0, -2.0], [-2.0, 14.0]]),
        ([40.2, 200.7], [[12.0, -5.0], [-5.0, 25.0]], [-4.0, 3.0], [37.5, 204.0], [[20.0, 3.0], [3.0, 6.0]]),
        ([3.0, 250.0], [[50.0, 0.0], [0.0, 40.0]], [0.0, 0.0], [1.0, 253.0], [[16.0, 1.0], [1.0, 16.0]]),   # clipped by the border
    ]
    out = {}
    for i, (m1, c1, a1, m2, c2) in enumerate(cases):
        m1, c1, a1, m2, c2 = (torch.tensor(v) for v in (m1, c1, a1, m2, c2))
        p1 = BivariateSkewNormal.pdf(grid, m1, c1, a1)
        p2 = torch.exp(MultivariateNormal(m2, c2, validate_args=False).log_prob(grid))
        p2b = BivariateNormal.pdf(grid, m2, c2)
        p = p1 * p2
        p = p / torch.sum(p)
        out.update({f"c{i}_mu1": npy(m1), f"c{i}_cov1": npy(c1), f"c{i}_alpha1": npy(a1), f"c{i}_mu2": npy(m2),
                    f"c{i}_cov2": npy(c2), f"c{i}_p1": npy(p1), f"c{i}_p2": npy(p2), f"c{i}_p2_bn": npy(p2b),
                    f"c{i}_p": npy(p)})
    # rvs_fast: MultivariateNormal.sample = scale_tril @ randn under the global generator
    m, c, a = torch.tensor([100.0, 150.0]), torch.tensor([[10.0, -5.0], [-5.0, 12.0]]), torch.tensor([4.0, -2.0])
    torch.manual_seed(11)
    xs = BivariateSkewNormal.rvs_fast(m, c, a, size=(64,))
    torch.manual_seed(11)
    eps = torch.randn(64, 3)
    out.update({"rvs_mu": npy(m), "rvs_cov": npy(c), "rvs_alpha": npy(a), "rvs_eps": npy(eps), "rvs_x": npy(xs)})
    # two-instant PSM conditional, fixed (file) model and re-centred model
    seq = np.load(REF / "camus-cont_sequence_psm_11_no_std.npy", allow_pickle=True).item()
    smu, sQ = torch.tensor(seq["mu"], dtype=torch.float), torch.tensor(seq["Q"], dtype=torch.float)
    xv = torch.tensor(np.asarray(seq["X_val"])[7], dtype=torch.float)
    for first in (0, 1):
        sg = torch.zeros(84)
        sl = slice(0, 42) if first == 0 else slice(42, 84)
        sg[sl] = xv[sl] + 1.5
        idx = index_to_flat(list(range(21)) if first == 0 else list(range(21, 42)))
        mu_c, cov_c = posterior_shape_model(sg.reshape(-1, 1), idx, smu, sQ, sigma2=1)
        out[f"seq{first}_sg"] = npy(sg)
        out[f"seq{first}_mu_c"] = npy(mu_c.squeeze())
        out[f"seq{first}_cov_blocks"] = npy(torch.stack([cov_c[2 * i:2 * i + 2, 2 * i:2 * i + 2] for i in range(42)]))
        pm, pQ = pca(torch.tensor(np.asarray(seq["X_train"]), dtype=torch.float), xv.reshape(-1, 1))
        mu_c, cov_c = posterior_shape_model(sg.reshape(-1, 1), idx, pm, pQ, sigma2=1)
        out[f"seqpca{first}_mu_c"] = npy(mu_c.squeeze())
        out[f"seqpca{first}_cov_blocks"] = npy(torch.stack([cov_c[2 * i:2 * i + 2, 2 * i:2 * i + 2] for i in range(42)]))
    out["seq_xv"] = npy(xv)
    np.savez_compressed(OUT / "skew_grid.npz", **out)


def gen_umap():
    """projected_uncertainty (utils/uncertainty_projection.py:17-129, importable) and the distribution marginals it calls,
    on LV-like contours: the per-landmark normal direction, projected standard deviation and projected skewness that
    skew_umap / uncertainty_map build their maps from (those two modules need scikit-image and are not importable)."""
    from contour_uncertainty.utils.uncertainty_projection import projected_uncertainty
    g = torch.Generator().manual_seed(11)
    out = {}
    for case in range(4):
        k = 21
        t = np.linspace(0.0, np.pi, k)
        rng = np.random.default_rng(case)
        mu = np.stack([128 + (50 + 10 * case) * np.cos(t), 170 - (80 + 5 * case) * np.sin(t)], -1) + rng.normal(size=(k, 2))
        mu = mu.astype(np.float32)          # the predict step hands float32 arrays over (aleatoric_skew.py:70-90)
        cov = rand_spd(k, g, 2.0, 60.0).numpy().astype(np.float32)
        alpha = (torch.randn(k, 2, generator=g) * 2.0).numpy().astype(np.float32)
        out[f"c{case}_mu"], out[f"c{case}_cov"], out[f"c{case}_alpha"] = mu, cov, alpha
        for lc in (False, True):
            u, v, a = projected_uncertainty(mu, cov, alpha.copy(), all=True, linear_close=lc)
            out[f"c{case}_lc{int(lc)}_u"], out[f"c{case}_lc{int(lc)}_v"] = np.asarray(u, dtype=np.float64), np.asarray(v)
            out[f"c{case}_lc{int(lc)}_a"] = np.asarray([float(x) for x in a])
        u, v = projected_uncertainty(mu, cov, all=True)
        out[f"c{case}_gauss_u"], out[f"c{case}_gauss_v"] = np.asarray([float(x) for x in u]), np.asarray(v)
        u, v = projected_uncertainty(mu, cov)
        out[f"c{case}_ends_u"] = np.asarray([float(x) for x in u])
    np.savez_compressed(OUT / "umap_projection.npz", **out)


def gen_skew_mode():
    """BivariateSkewNormal.mode and the univariate summaries it is built from (bivariateskewnormal.py:73-82,195-219)."""
    from contour_uncertainty.distributions import bivariateskewnormal as B
    g = torch.Generator().manual_seed(21)
    m = 12
    mu = torch.rand(m, 2, generator=g) * 200 + 20
    cov = rand_spd(m, g, 2.0, 80.0)
    alpha = torch.randn(m, 2, generator=g) * 3
    alpha[0] = torch.tensor([5.0, 0.0])                      # the reference's own check_bivariate_mode() constants
    cov[0] = torch.tensor([[10.0, -5.0], [-5.0, 10.0]])
    mu[0] = torch.tensor([100.0, 150.0])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        modes = torch.stack([B.BivariateSkewNormal.mode(mu[i], cov[i], alpha[i]) for i in range(m)])
    a1 = torch.tensor([-8.0, -3.0, -1.0, -0.2, 0.3, 1.0, 3.0, 8.0, 25.0])
    out = {"mu": npy(mu), "cov": npy(cov), "alpha": npy(alpha), "mode": npy(modes), "a1": npy(a1),
           "delta": npy(B.delta(a1)), "skewness": npy(B.skewness(a1)), "m0": npy(B.m0(a1)),
           "univariate_mode": npy(B.univariate_mode(torch.tensor(3.0), torch.tensor(2.0), a1))}
    np.savez_compressed(OUT / "skew_mode.npz", **out)


def gen_drop():
    """Which ConvLayers carry a Dropout2d when task.model.drop_block=True (reference unet2.py:129-136,302; layers.py
    196-202): read off the instantiated reference modules for the 6-stage (config c1) and 8-stage (unet2.yaml) nets."""
    import json
    from contour_uncertainty.models.nnUnet.unet2 import UNet
    res = {}
    for n in (4, 6, 8):
        net = UNet((1, 0, 0), (21, 0, 0), [256, 256], [[3, 3]] * n, [[1, 1]] + [[2, 2]] * (n - 1), drop_block=True)
        res[str(n)] = sorted(name for name, m in net.named_modules() if getattr(m, "use_drop_block", False)
                             and hasattr(m, "conv") and not hasattr(m, "conv1"))
    (OUT / "drop_block_layers.json").write_text(json.dumps(res, indent=1) + "\n")


def gen_vital_unet():
    """The `vital` U-Net (vital/vital/models/segmentation/unet.py, loaded by file path: the package's __init__ needs
    dotenv): init_channels 32, K = 5, N = 3, 64 x 64, train mode: logits, running statistics after the forward,
    per-parameter gradient statistics; then the eval-mode logits with the updated statistics."""
    import importlib.util
    from oracle import vital_unet as OV
    spec = importlib.util.spec_from_file_location("ref_vital_unet", REF / "vital/vital/models/segmentation/unet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(23)
    sd = OV.init_state(1, 5, 32, g)
    net = mod.UNet((1, 64, 64), (5, 64, 64), init_channels=32)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    net.train()
    x = torch.rand(3, 1, 64, 64, generator=g)
    logits = net(x)
    gl = torch.randn(logits.shape, generator=g)
    (logits * gl).sum().backward()
    out = {"x": npy(x), "logits": npy(logits), "g_logits": npy(gl)}
    names, gstats, ghead = [], [], []
    for name, p in net.named_parameters():
        names.append(name)
        gstats.append(stats(p.grad))
        ghead.append(npy(p.grad.flatten()[:8]))
    out["grad_names"] = np.array(names)
    out["grad_stats"] = np.stack(gstats)
    out["grad_head"] = np.stack([np.pad(h, (0, 8 - len(h))) for h in ghead])
    new = net.state_dict()
    for k in ("layer1.net.1", "layer6.net.1.net.5", "layer11.conv.net.5"):
        out[f"rm:{k}"] = npy(new[f"{k}.running_mean"])
        out[f"rv:{k}"] = npy(new[f"{k}.running_var"])
    net.eval()
    with torch.no_grad():
        out["logits_eval"] = npy(net(x))
    np.savez_compressed(OUT / "vital_unet.npz", **out)


def gen_ckpt():
    """A Lightning-1.8-layout checkpoint of the REFERENCE's module structure.

    The reference tasks hold ``self.model = UNet(...)`` and, for dsnt-skew, ``self.skew_block = model.confidence_net(2K)``
    (reference task/regression/dsnt/dsnt_skew.py:34); Lightning itself is not importable here, so the LightningModule shell is
    a plain ``nn.Module`` with those two attributes -- the reference's own ``UNet`` / ``ConfidenceNet`` classes inside, hence
    the reference's ``state_dict`` names, order and shapes -- stepped once with ``torch.optim.Adam`` (reference
    vital/vital/config/task/optim/adam.yaml).  Written:
      * ref_ckpt_tiny.ckpt      2-stage dsnt-al task (109 k parameters): the real file, state_dict + optimizer_states;
      * ref_ckpt_structure.json names / shapes / dtypes of the 8-stage dsnt-skew task's state_dict and the layout of its Adam
        state (too large to commit as tensors)."""
    import json
    from contour_uncertainty.models.nnUnet.unet2 import UNet

    class Shell(torch.nn.Module):
        def __init__(self, n_stages, skew):
            super().__init__()
            self.model = UNet((1, 0, 0), (21, 0, 0), [256, 256], [[3, 3]] * n_stages, [[1, 1]] + [[2, 2]] * (n_stages - 1),
                              bottleneck_out=skew)
            if skew:
                self.skew_block = self.model.confidence_net(42)

    def one_step(shell, size):
        opt = torch.optim.Adam(shell.parameters(), lr=1e-3, weight_decay=1e-3)
        x = torch.rand(2, 1, size, size, generator=torch.Generator().manual_seed(3))
        out = shell.model(x)
        loss = out[0].square().mean() + (shell.skew_block(out[1]).square().mean() if hasattr(shell, "skew_block") else 0.0) \
            if isinstance(out, tuple) else out.square().mean()
        loss.backward()
        opt.step()
        return opt

    torch.manual_seed(5)
    tiny = Shell(2, False)
    opt = one_step(tiny, 16)
    ckpt = {"epoch": 3, "global_step": 17, "pytorch-lightning_version": "1.8.0", "state_dict": tiny.state_dict(), "loops": {},
            "callbacks": {}, "optimizer_states": [opt.state_dict()], "lr_schedulers": [], "hparams_name": "kwargs",
            "hyper_parameters": {"covar": True, "mse_weight": 1, "log_penalty_weight": 1, "t_a": 25, "t_e": 1}}
    torch.save(ckpt, str(OUT / "ref_ckpt_tiny.ckpt"))
    x = torch.rand(1, 1, 16, 16, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        np.savez_compressed(OUT / "ref_ckpt_tiny_io.npz", x=npy(x), logits=npy(tiny.model(x)))
    torch.manual_seed(6)
    full = Shell(8, True)
    opt = one_step(full, 256)
    osd = opt.state_dict()
    structure = {
        "state_dict": [[k, list(v.shape), str(v.dtype)] for k, v in full.state_dict().items()],
        "optimizer_param_groups": [{k: (v if k != "params" else len(v)) for k, v in g.items()} for g in osd["param_groups"]],
        "optimizer_state_keys": sorted({k for st in osd["state"].values() for k in st}),
        "optimizer_state_count": len(osd["state"]),           # parameters that received a gradient
        "optimizer_state_ids": sorted(int(k) for k in osd["state"]),
        "parameter_order": [n for n, _ in full.named_parameters()],
        "step_dtype": str(next(iter(osd["state"].values()))["step"].dtype),
    }
    (OUT / "ref_ckpt_structure.json").write_text(json.dumps(structure, indent=0))


def gen_clinical():
    """aleatoric / epistemic split of a Monte-Carlo metric (reference results/clinical/utils.py: importable, NumPy only)"""
    from contour_uncertainty.results.clinical.utils import aleatoric_epistemic_uncertainty
    rng = np.random.default_rng(12)
    out = {}
    for i, (te, ta, nan_frac) in enumerate(((1, 25, 0.0), (5, 40, 0.1), (3, 1024, 0.02))):
        mc = rng.normal(0.45, 0.08, size=(te, ta))
        mc[rng.random(mc.shape) < nan_frac] = np.nan
        out[f"mc{i}"] = mc
        out[f"res{i}"] = np.array(aleatoric_epistemic_uncertainty(mc), dtype=np.float64)
    np.savez_compressed(OUT / "clinical.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["dsnt", "nll", "unet_small", "unet_full", "step", "psm", "skew_grid", "umap", "drop", "skew_mode",
                             "vital_unet", "ckpt", "clinical"]
    for w in which:
        print("generating", w, flush=True)
        {"dsnt": gen_dsnt, "nll": gen_nll, "unet_small": gen_unet_small, "unet_full": gen_unet_full,
         "step": gen_step, "psm": gen_psm, "skew_grid": gen_skew_grid, "umap": gen_umap, "drop": gen_drop, "skew_mode": gen_skew_mode,
         "vital_unet": gen_vital_unet, "ckpt": gen_ckpt, "clinical": gen_clinical}[w]()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
