"""End-to-end GPU parity: the drop-in UNet / DSNT tasks (HIP kernels through the C ABI) against the golden vectors
generated from the imported reference, and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import head as OH
from oracle import unet as OU
from oracle.step import OracleTask, synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def model_cfg(n_stages, dtype):
    return {"_target_": "contour_uncertainty.models.nnUnet.unet2.UNet", "kernels": [[3, 3]] * n_stages,
            "strides": [[1, 1]] + [[2, 2]] * (n_stages - 1), "patch_size": [256, 256], "drop_block": False,
            "deep_supervision": False, "compute_dtype": dtype}


def make_task(kind, n_stages, size, dtype, **kw):
    from contour_uncertainty._compat import DataParameters
    from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
    from contour_uncertainty.task.regression.dsnt.dsnt_skew import DSNTSkew
    cls = DSNTSkew if kind == "dsnt-skew" else DSNTAleatoric
    optim = {"_target_": "torch.optim.Adam", "lr": 1e-3, "weight_decay": 1e-3}
    return cls(model=model_cfg(n_stages, dtype), optim=optim, choices={},
               data_params=DataParameters((1, size, size), (21, 2), [0, 1]), psm_path="camus-cont_psm_11_no_std.npy",
               seq_psm_path="camus-cont_sequence_psm_11_no_std.npy", t_a=25, t_e=1, covar=True, **kw)


def small_state():
    spec = OU.UNetSpec(in_channels=1, num_classes=5, strides=(1, 2, 2, 2))
    g = torch.Generator().manual_seed(11)
    sd = OU.init_unet_state(spec, g)
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    x = torch.rand(2, 1, 32, 32, generator=g)
    return spec, sd, x


def test_unet_small_fwd_bwd_vs_reference_golden(golden_dir):
    """4-stage net, f32 parity mode: logits, bottleneck and every parameter gradient vs the imported reference."""
    from contour_uncertainty.models.nnUnet.unet2 import UNet
    g = np.load(golden_dir / "unet_small.npz")
    spec, sd, x = small_state()
    net = UNet((1, 32, 32), (5, 1, 32), [256, 256], [[3, 3]] * 4, [[1, 1]] + [[2, 2]] * 3, bottleneck_out=True,
               compute_dtype="f32")
    res = net.load_state_dict(sd, strict=True)            # reference names/shapes, strict
    assert not res.missing_keys and not res.unexpected_keys
    net = net.to(DEV)
    logits, bott = net(x.to(DEV))
    assert torch.allclose(logits.cpu(), T(g["logits"]), rtol=1e-4, atol=2e-5)
    assert torch.allclose(bott.cpu(), T(g["bottleneck"]), rtol=1e-4, atol=2e-5)
    ((logits * T(g["g_logits"]).to(DEV)).sum() + (bott * T(g["g_bott"]).to(DEV)).sum()).backward()
    params = dict(net.named_parameters())
    for i, name in enumerate(str(n) for n in g["grad_names"]):
        gr = params[name].grad
        assert gr is not None, name
        ref = g["grad_stats"][i]
        l2 = float(gr.double().pow(2).sum().sqrt())
        if name.endswith("conv.bias") and "output" not in name:
            continue   # bias in front of InstanceNorm: analytically zero, pure rounding noise in the reference too
        assert abs(l2 - ref[2]) <= 5e-4 * max(ref[2], 1e-6), (name, l2, ref[2])
        head = gr.flatten()[:8].cpu().numpy()
        assert np.allclose(head, g["grad_head"][i][: len(head)], rtol=5e-3, atol=2e-4 * max(ref[2], 1e-6)), name
    for name in (str(n) for n in g["no_grad_names"]):
        assert params[name].grad is None


@pytest.mark.parametrize("kind", ["dsnt-skew", "dsnt-al"])
def test_train_steps_vs_reference_golden(golden_dir, kind):
    """BASELINE config c1 shape (6 stages, 64x64, batch 2), f32 parity mode: two full training steps
    (forward, backward, fused Adam) reproduce the reference-composed step's logged values and updated weights."""
    g = np.load(golden_dir / "train_step.npz")
    task = make_task(kind, 6, 64, "f32")
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    gen = torch.Generator().manual_seed(0)
    task.model.load_state_dict(OU.init_unet_state(spec, gen), strict=True)
    if kind == "dsnt-skew":
        task.skew_block.load_state_dict(OU.init_confidence_state(42, gen), strict=True)
    task = task.to(DEV)
    opt = task.configure_optimizers()["optimizer"]
    img, contour = synthetic_batch(2, 64, 21, seed=1234)
    batch = {"img": img.to(DEV), "contour": contour.to(DEV)}
    keys = ["loss", "distance_loss", "loss_term1", "loss_term2", "loss_term3", "alpha_norm"]
    for it in range(2):
        opt.zero_grad(set_to_none=True)
        out = task.training_step(batch, it)
        assert set(out.keys()) >= {"loss", "train/loss", "train/distance_loss", "train/loss_term1", "train/loss_term2"}
        out["loss"].backward()
        opt.step()
        ref = g[f"{kind}_logs"][it]
        for j, r in enumerate(ref):
            v = float(out[f"train/{keys[j]}"])
            assert abs(v - r) <= 3e-4 * max(1.0, abs(r)), (it, keys[j], v, r)
    # Adam moves every weight by ~lr per step whatever the gradient's size, so a LeakyReLU kink decided differently by
    # rounding (|pre-activation| ~ 1e-7) shows up as a fraction of 2*lr = 2e-3: compare at 25 % of the maximum move
    sd = task.model.state_dict()
    assert torch.allclose(sd["output_block.conv.weight"].cpu(), T(g[f"{kind}_w_out"]), rtol=0, atol=5e-4)
    assert torch.allclose(sd["input_block.conv1.conv.weight"].cpu(), T(g[f"{kind}_w_in"]), rtol=0, atol=5e-4)


@pytest.mark.parametrize("kind", ["dsnt-skew", "dsnt-al"])
def test_predict_mu_sigma_alpha_vs_reference_golden(golden_dir, kind):
    """mu / Sigma (/ alpha) of predict_on_batch within 1e-4 relative of the reference PyTorch-CPU path (north_star)."""
    g = np.load(golden_dir / "train_step.npz")
    task = make_task(kind, 6, 64, "f32")
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    gen = torch.Generator().manual_seed(0)
    task.model.load_state_dict(OU.init_unet_state(spec, gen), strict=True)
    if kind == "dsnt-skew":
        task.skew_block.load_state_dict(OU.init_confidence_state(42, gen), strict=True)
    task = task.to(DEV)
    img, _ = synthetic_batch(2, 64, 21, seed=1234)
    out = task.predict_on_batch(img.to(DEV), task.model)
    mu_ref, sig_ref = T(g[f"{kind}_mu0"]), T(g[f"{kind}_sigma0"])
    assert float((out[0].cpu() - mu_ref).abs().max() / mu_ref.abs().max()) < 1e-4
    assert float((out[1].cpu() - sig_ref).abs().max() / sig_ref.abs().max()) < 1e-4
    S, cov = task.predict(img.to(DEV))[:2]
    assert S.shape == (2, 1, 21, 2) and cov.shape == (2, 1, 21, 2, 2) and not S.is_cuda


def test_full_size_forward_vs_reference_golden(golden_dir):
    """The real 8-stage unet2.yaml network at 256x256 (N=1), f32 parity mode, vs the imported reference."""
    from contour_uncertainty.models.nnUnet.unet2 import ConfidenceNet, UNet
    from cu_hip.head import dsnt_moments
    from cu_hip import ops
    g = np.load(golden_dir / "unet_full.npz")
    spec = OU.UNetSpec()
    gen = torch.Generator().manual_seed(0)
    sd = OU.init_unet_state(spec, gen)
    ssd = OU.init_confidence_state(42, gen)
    x = torch.rand(1, 1, 256, 256, generator=gen)
    net = UNet((1, 256, 256), (21, 1, 256), [256, 256], [[3, 3]] * 8, [[1, 1]] + [[2, 2]] * 7, bottleneck_out=True,
               compute_dtype="f32")
    net.load_state_dict(sd, strict=True)
    head = ConfidenceNet(42, compute_dtype="f32")
    head.load_state_dict(ssd, strict=True)
    net, head = net.to(DEV), head.to(DEV)
    with torch.no_grad():
        logits, bott = net(x.to(DEV))
        a = head(bott)
        mu, sigma, aux = ops.dsnt_head_fwd(logits, True)
    # the bottleneck is 16 InstanceNorms deep and normalises over 2x2 = 4 values: ill-conditioned, so it (and the skew
    # head fed by it) gets an absolute tolerance; the head outputs below are what north_star bounds at 1e-4
    assert torch.allclose(bott.cpu(), T(g["bottleneck"]), rtol=5e-3, atol=2e-3)
    assert torch.allclose(a.cpu(), T(g["alpha_raw"]), rtol=5e-3, atol=2e-3)
    assert torch.allclose(logits[0, :, 128, :].cpu(), T(g["logits_row"]), rtol=5e-3, atol=1e-3)
    coords_px = 0.5 * ((T(g["coords"]) + 1) * 256 - 1)
    assert float((mu.cpu() - coords_px).abs().max() / coords_px.abs().max()) < 1e-4
    var_px = T(g["var"]) * 128.0 ** 2
    assert float((sigma[..., :2].cpu() - var_px).abs().max() / var_px.abs().max()) < 1e-4


def test_bf16_step_tracks_f32(golden_dir):
    """Production mode (bf16 MFMA, f32 accumulate): same step as above; the NLL and the gradient direction must track
    the f32 parity mode ("contour-NLL vs ref" of BASELINE.json's metric)."""
    res = {}
    for dtype in ("f32", "bf16"):
        task = make_task("dsnt-skew", 6, 64, dtype)
        spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
        gen = torch.Generator().manual_seed(0)
        task.model.load_state_dict(OU.init_unet_state(spec, gen), strict=True)
        task.skew_block.load_state_dict(OU.init_confidence_state(42, gen), strict=True)
        task = task.to(DEV)
        img, contour = synthetic_batch(4, 64, 21, seed=1234)
        out = task.training_step({"img": img.to(DEV), "contour": contour.to(DEV)}, 0)
        out["loss"].backward()
        flat, grad = task.model.flat_params()
        res[dtype] = (float(out["loss"]), grad.clone())
    lf, gf = res["f32"]
    lb, gb = res["bf16"]
    assert abs(lb - lf) < 0.05 * abs(lf), (lb, lf)
    cos = float(torch.dot(gf, gb) / (gf.norm() * gb.norm()))
    assert cos > 0.98, cos


def test_oracle_vs_hip_larger_batch():
    """Oracle (CPU) and HIP f32 path on the same seeded inputs at a size the oracle finishes in seconds: 6 stages,
    64x64, batch 6 (not a multiple of the tile's image count), dsnt-al2 branch (covar=True) and covar=False."""
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    img, contour = synthetic_batch(6, 64, 21, seed=99)
    for covar in (True, False):
        ot = OracleTask(spec, task="dsnt-al", covar=covar, seed=3)
        ref = ot.forward_loss(img, contour)
        ref["loss"].backward()
        task = make_task("dsnt-al", 6, 64, "f32")
        task.hparams.covar = covar
        task.model.load_state_dict({k: v.detach() for k, v in ot.sd.items()}, strict=True)
        task = task.to(DEV)
        out = task._shared_step({"img": img.to(DEV), "contour": contour.to(DEV)}, 0)
        out["loss"].backward()
        for k in ("loss", "distance_loss", "loss_term1", "loss_term2"):
            assert abs(float(out[k]) - float(ref[k])) <= 2e-4 * max(1.0, abs(float(ref[k]))), (covar, k)
        params = dict(task.model.named_parameters())
        for name in ("output_block.conv.weight", "upsamples.0.transp_conv.weight", "bottleneck.conv1.conv.weight",
                     "downsamples.0.conv1.conv.weight", "input_block.conv1.conv.weight",
                     "upsamples.4.conv_block.conv1.conv.weight", "input_block.conv2.norm.weight"):
            a, b = params[name].grad.cpu(), ot.sd[name].grad
            err = float((a - b).norm() / b.norm())
            assert err < 2e-3, (covar, name, err)
