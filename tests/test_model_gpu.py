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


def make_task(kind, n_stages, size, dtype, t_e=1, drop_block=False, **kw):
    from contour_uncertainty._compat import DataParameters
    from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
    from contour_uncertainty.task.regression.dsnt.dsnt_skew import DSNTSkew
    cls = DSNTSkew if kind == "dsnt-skew" else DSNTAleatoric
    optim = {"_target_": "torch.optim.Adam", "lr": 1e-3, "weight_decay": 1e-3}
    cfg = dict(model_cfg(n_stages, dtype), drop_block=drop_block)
    return cls(model=cfg, optim=optim, choices={},
               data_params=DataParameters((1, size, size), (21, 2), [0, 1]), psm_path="camus-cont_psm_11_no_std.npy",
               seq_psm_path="camus-cont_sequence_psm_11_no_std.npy", t_a=25, t_e=t_e, covar=True, **kw)


def hip_kink_masks(model):
    """LeakyReLU sign pattern the device used in its last forward (from the materialised activations), per conv layer."""
    ctx = model.engine._last_ctx
    return {prefix: (rec.out.a.float() > 0).permute(0, 3, 1, 2).cpu() for prefix, rec in ctx.convs.items()}


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
    (forward, backward, fused Adam).
      * every logged value of both steps vs the reference-composed step (golden, 3e-4);
      * the updated weights vs the CPU oracle stepping in lock-step with the device's LeakyReLU sign pattern
        (LeakyReLU' is discontinuous at 0: a pre-activation of +-1e-7 decided differently by two correct
        implementations changes that element's gradient 100x, and Adam turns any gradient change into an O(lr) move);
      * the updated weights vs the golden reference within half of the largest possible Adam move (2 * lr)."""
    g = np.load(golden_dir / "train_step.npz")
    task = make_task(kind, 6, 64, "f32")
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    gen = torch.Generator().manual_seed(0)
    task.model.load_state_dict(OU.init_unet_state(spec, gen), strict=True)
    if kind == "dsnt-skew":
        task.skew_block.load_state_dict(OU.init_confidence_state(42, gen), strict=True)
    task = task.to(DEV)
    task.model.engine.debug = {}
    ot = OracleTask(spec, task=kind, seed=0)
    opt = task.configure_optimizers()["optimizer"]
    img, contour = synthetic_batch(2, 64, 21, seed=1234)
    batch = {"img": img.to(DEV), "contour": contour.to(DEV)}
    keys = ["loss", "distance_loss", "loss_term1", "loss_term2", "loss_term3", "alpha_norm"]
    n_flip = 0
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
        masks = hip_kink_masks(task.model)
        taps = {}
        OU.unet_forward({k: v.detach() for k, v in ot.sd.items()}, img, spec, taps=taps)
        n_flip += sum(int(((taps[f"{p}:a"] > 0) != m).sum()) for p, m in masks.items())
        ot.train_step(img, contour, masks=masks)
    n_act = sum(m.numel() for m in masks.values())
    # observed ~7e-5 of the decisions: pre-activations within rounding noise of 0 (the 2x2 / 4x4 InstanceNorms
    # amplify 1e-7 relative differences of the conv outputs to ~1e-4)
    assert n_flip <= 5e-4 * 2 * n_act, f"{n_flip} sign decisions differ out of {2 * n_act}"
    sd = task.model.state_dict()
    for name in ("output_block.conv.weight", "input_block.conv1.conv.weight", "bottleneck.conv1.conv.weight",
                 "upsamples.2.transp_conv.weight", "downsamples.1.conv2.norm.weight"):
        # Adam's update m/sqrt(v) is scale-free: an element whose gradient is at rounding-noise level moves by O(lr)
        # in a noise-decided direction, so a handful of elements may differ by up to 2*lr; all others must agree tightly
        a_, b_ = sd[name].cpu(), ot.sd[name].detach()
        bad = ((a_ - b_).abs() > 5e-5 + 1e-3 * b_.abs()).float().mean()
        assert float(bad) < 0.01 and float((a_ - b_).abs().max()) <= 2.2e-3, (name, float(bad))
    for name, key in (("output_block.conv.weight", "w_out"), ("input_block.conv1.conv.weight", "w_in")):
        diff = (sd[name].cpu() - T(g[f"{kind}_{key}"])).abs()
        assert float(diff.max()) <= 2.2e-3 and float((diff > 5e-4).float().mean()) < 0.05, (name, float(diff.max()))


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
    res = task.predict(img.to(DEV))
    S, cov = res[:2]
    assert S.shape == (2, 1, 21, 2) and cov.shape == (2, 1, 21, 2, 2) and not S.is_cuda
    if kind == "dsnt-skew":
        # alpha of predict_on_batch: the reference's raw head output with alpha_y negated (dsnt_skew.py:164), 1e-4
        a_ref = T(g["dsnt-skew_alpha0_predict"])
        assert float((out[2].cpu() - a_ref).abs().max() / a_ref.abs().max()) < 1e-4
        assert torch.equal(T(g["dsnt-skew_alpha0"])[..., 0], a_ref[..., 0])
        assert torch.equal(T(g["dsnt-skew_alpha0"])[..., 1], -a_ref[..., 1])
        assert res[2].shape == (2, 1, 21, 2)
        assert float((res[2][:, 0] - a_ref).abs().max() / a_ref.abs().max()) < 1e-4


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
    # Tolerances (measured, not guessed).  The reference algorithm itself, run on the CPU in
    # float32 and in float64 on these weights, differs by 1.0e-4 absolute at the bottleneck (16 InstanceNorms deep, the last
    # ones over 2x2 = 4 values: ill-conditioned) and by 6.1e-5 in the logits; the HIP f32 path measured against the
    # reference's float32 golden (tools/full_size_err.py, profiles/r03_full_size_err.txt): bottleneck 4.7e-4 absolute
    # (|max| 1.73), alpha 5.3e-7 (|max| 0.058), logits 9.3e-5 (|max| 4.6).  The bounds leave 2x for the run-to-run noise of
    # the f32 atomics in the statistics; the head outputs below are what north_star bounds at 1e-4.
    assert torch.allclose(bott.cpu(), T(g["bottleneck"]), rtol=1e-3, atol=8e-4)
    assert torch.allclose(a.cpu(), T(g["alpha_raw"]), rtol=1e-3, atol=5e-6)
    assert torch.allclose(logits[0, :, 128, :].cpu(), T(g["logits_row"]), rtol=3e-4, atol=2e-4)
    coords_px = 0.5 * ((T(g["coords"]) + 1) * 256 - 1)
    assert float((mu.cpu() - coords_px).abs().max() / coords_px.abs().max()) < 1e-4
    var_px = T(g["var"]) * 128.0 ** 2
    assert float((sigma[..., :2].cpu() - var_px).abs().max() / var_px.abs().max()) < 1e-4


def test_bf16_step_tracks_f32(golden_dir):
    """Production mode (bf16 operands, activations and activation gradients; f32 accumulate, statistics, weights):
    "contour-NLL vs ref" of BASELINE.json's metric.  The loss must agree with the f32 parity mode to 1e-3, and the
    gradient noise must be what bf16 storage itself causes: per layer, the device's error w.r.t. the f32 gradient is
    compared with a CPU simulation that only rounds the same tensors to bf16 (oracle.unet._RoundBf16).  (At random
    init this network amplifies rounding noise strongly through its 2x2 / 4x4 InstanceNorms: ~3 % at the output layer,
    ~75 % at the first layer -- for the simulation and for the kernels alike.)"""
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    img, contour = synthetic_batch(4, 64, 21, seed=1234)
    ot = OracleTask(spec, task="dsnt-skew", seed=0)
    ref = ot.forward_loss(img, contour)
    ref["loss"].backward()
    g32 = {k: v.grad.clone() for k, v in ot.sd.items() if v.grad is not None}
    ot2 = OracleTask(spec, task="dsnt-skew", seed=0)
    sim = ot2.forward_loss(img, contour, round_bf16=True)
    sim["loss"].backward()
    task = make_task("dsnt-skew", 6, 64, "bf16")
    task.model.load_state_dict({k: v.detach() for k, v in ot.sd.items()}, strict=True)
    task.skew_block.load_state_dict({k: v.detach() for k, v in ot.skew_sd.items()}, strict=True)
    task = task.to(DEV)
    out = task.training_step({"img": img.to(DEV), "contour": contour.to(DEV)}, 0)
    out["loss"].backward()
    assert abs(float(out["loss"]) - float(ref["loss"])) < 1e-3 * abs(float(ref["loss"]))
    params = dict(task.model.named_parameters())
    for name in ("output_block.conv.weight", "upsamples.4.conv_block.conv2.conv.weight",
                 "upsamples.3.conv_block.conv1.conv.weight", "upsamples.1.transp_conv.weight",
                 "bottleneck.conv1.conv.weight", "downsamples.1.conv1.conv.weight", "input_block.conv2.conv.weight"):
        truth = g32[name]
        e_hip = float((params[name].grad.cpu() - truth).norm() / truth.norm())
        e_sim = float((ot2.sd[name].grad - truth).norm() / truth.norm())
        assert e_hip <= 1.6 * e_sim + 0.02, (name, e_hip, e_sim)


def test_oracle_vs_hip_larger_batch():
    """Oracle (CPU) and HIP f32 path on the same seeded inputs at a size the oracle finishes in seconds: 6 stages,
    64x64, batch 6 (not a multiple of the tile's image count), dsnt-al2 branch (covar=True) and covar=False.
    Every parameter gradient is compared; the oracle's backward is given the device's LeakyReLU sign pattern (see
    test_train_steps_vs_reference_golden) and the number of differing sign decisions is bounded separately."""
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    img, contour = synthetic_batch(6, 64, 21, seed=99)
    for covar in (True, False):
        ot = OracleTask(spec, task="dsnt-al", covar=covar, seed=3)
        task = make_task("dsnt-al", 6, 64, "f32")
        task.hparams.covar = covar
        task.model.load_state_dict({k: v.detach() for k, v in ot.sd.items()}, strict=True)
        task = task.to(DEV)
        task.model.engine.debug = {}
        out = task._shared_step({"img": img.to(DEV), "contour": contour.to(DEV)}, 0)
        out["loss"].backward()
        masks = hip_kink_masks(task.model)
        taps = {}
        ref = ot.forward_loss(img, contour, masks=masks)
        ref["loss"].backward()
        for k in ("loss", "distance_loss", "loss_term1", "loss_term2"):
            assert abs(float(out[k]) - float(ref[k])) <= 2e-4 * max(1.0, abs(float(ref[k]))), (covar, k)
        params = dict(task.model.named_parameters())
        worst = 0.0
        for name, p in params.items():
            if p.grad is None or (name.endswith("conv.bias")):
                continue
            b = ot.sd[name].grad
            err = float((p.grad.cpu() - b).norm() / b.norm())
            worst = max(worst, err)
            assert err < 5e-4, (covar, name, err)


def test_drop_block_matches_oracle_with_shared_masks():
    """task.model.drop_block=True (tmi_scripts/train.sh:9): Dropout2d between conv and norm in the LAST downsample block
    and the bottleneck (reference unet2.py:302 with in_channels = filters[:-1]; the flagged layer names are pinned by
    tests/golden/drop_block_layers.json, generated from the imported reference), training mode only.  Same channel masks
    on both sides -> same loss and gradients."""
    from contour_uncertainty.models.nnUnet.unet2 import UNet
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    gen = torch.Generator().manual_seed(4)
    sd = OU.init_unet_state(spec, gen)
    img, contour = synthetic_batch(3, 64, 21, seed=5)
    net = UNet((1, 64, 64), (21, 1, 64), [256, 256], [[3, 3]] * 6, [[1, 1]] + [[2, 2]] * 5, drop_block=True,
               compute_dtype="f32")
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    assert sorted(net.engine.drop_layers) == ["bottleneck.conv1", "bottleneck.conv2", "downsamples.3.conv1",
                                              "downsamples.3.conv2"]
    masks = {}

    def mask_fn(prefix, n, c, device):
        import zlib
        g = torch.Generator().manual_seed(zlib.crc32(prefix.encode()) % 1000)      # (str hashes change from run to run)
        masks[prefix] = (torch.rand(n, c, generator=g) >= 0.5).float() * 2.0
        return masks[prefix].to(device)

    net.engine.drop_mask_fn = mask_fn
    net.engine.debug = {}
    net.train()
    logits = net(img.to(DEV))
    from cu_hip.head import dsnt_nll
    logs, _, _ = dsnt_nll(logits, contour.to(DEV))
    logs["loss"].backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kinks = hip_kink_masks(net)
    ref_logits = OU.unet_forward(sdr, img, spec, masks=kinks, drop=masks)
    ref = OH.dsnt_al_loss(ref_logits, contour)
    ref["loss"].backward()
    assert abs(float(logs["loss"]) - float(ref["loss"])) < 2e-4 * abs(float(ref["loss"]))
    params = dict(net.named_parameters())
    for name in ("downsamples.3.conv1.conv.weight", "bottleneck.conv2.conv.weight", "downsamples.3.conv2.norm.weight",
                 "input_block.conv1.conv.weight", "upsamples.0.transp_conv.weight"):
        a, b = params[name].grad.cpu(), sdr[name].grad
        assert float((a - b).norm() / b.norm()) < 5e-4, name
    # eval mode: no dropout, different output
    net.eval()
    with torch.no_grad():
        ev = net(img.to(DEV))
    ref_eval = OU.unet_forward(sd, img, spec)
    assert torch.allclose(ev.cpu(), ref_eval, rtol=1e-3, atol=1e-3)


def test_mc_dropout_and_ensemble_predict(tmp_path):
    """t_e > 1 keeps Dropout2d active at predict time (reference uncertainty.py:71-75 + utils/mcdropout.py:89-137);
    ensemble_ckpt builds one network per checkpoint and predict() stacks them (uncertainty.py:55-70, dsnt_al.py:133-151)."""
    img, _ = synthetic_batch(2, 64, 21, seed=8)
    torch.manual_seed(0)
    t = make_task("dsnt-al", 6, 64, "f32", t_e=3, drop_block=True).to(DEV).eval()
    assert t.model.mc_dropout
    mu, cov = t.predict(img.to(DEV))
    assert mu.shape == (2, 3, 21, 2) and cov.shape == (2, 3, 21, 2, 2)
    assert not torch.allclose(mu[:, 0], mu[:, 1])            # different dropout masks per epistemic sample
    t1 = make_task("dsnt-al", 6, 64, "f32", t_e=1, drop_block=True).to(DEV).eval()
    a, _ = t1.predict(img.to(DEV))
    b, _ = t1.predict(img.to(DEV))
    assert torch.allclose(a, b, rtol=0, atol=1e-4) and a.shape == (2, 1, 21, 2)    # t_e = 1: dropout is off in eval mode
    with pytest.warns(UserWarning):
        make_task("dsnt-al", 6, 64, "f32", t_e=2, drop_block=False)
    # ensemble of two differently initialised networks
    singles, paths = [], []
    for i in range(2):
        torch.manual_seed(10 + i)
        m = make_task("dsnt-al", 6, 64, "f32").to(DEV).eval()
        singles.append(m.predict(img.to(DEV)))
        paths.append(tmp_path / f"m{i}.ckpt")
        m.save_checkpoint(paths[-1])
    ens = make_task("dsnt-al", 6, 64, "f32", ensemble_ckpt=[str(p) for p in paths]).to(DEV).eval()
    assert ens.ensembling and len(ens.model) == 2
    mu, cov = ens.predict(img.to(DEV))
    assert ens.hparams.t_e == 2 and mu.shape == (2, 2, 21, 2)
    for i in range(2):
        assert torch.allclose(mu[:, i], singles[i][0][:, 0], rtol=1e-5, atol=1e-4)
        assert torch.allclose(cov[:, i], singles[i][1][:, 0], rtol=1e-4, atol=1e-3)
    ens_dir = make_task("dsnt-al", 6, 64, "f32", ensemble_ckpt=str(tmp_path))
    assert len(ens_dir.model) == 2


@pytest.mark.parametrize("kind,seq", [("dsnt-al", False), ("dsnt-al", True), ("dsnt-skew", False), ("dsnt-skew", True)])
def test_predict_step_sampling_fanout(golden_dir, kind, seq):
    """predict() -> sample() -> _predict_step / predict_step on the batched GPU samplers (Gaussian,
    skew-normal, and the ED/ES sequence variants: the batch is then one pair).  Mask rasterisation is host code of the
    datamodule (8f): a stub stands in."""
    from contour_uncertainty._compat import ContourTags, Tags
    n, size, k, t_a = 2, 64, 21, 6
    img, contour = synthetic_batch(n, size, k, seed=3)
    torch.manual_seed(1)
    t = make_task(kind, 6, size, "f32", sequence_sampler=seq).to(DEV).eval()
    t.hparams.psm_path = str(golden_dir / "camus-cont_psm_11_no_std.npz")
    t.hparams.seq_psm_path = str(golden_dir / "camus-cont_sequence_psm_11_no_std.npz")
    t.hparams.t_a = t_a

    def to_mask(c, shape, labels, apply_argmax=True):
        m = np.zeros((1,) + tuple(shape), dtype=np.float32)
        ij = np.clip(np.round(c).astype(int), 0, shape[0] - 1)
        m[0, ij[:, 1], ij[:, 0]] = 1.0
        return m

    t.contour_to_mask_fn = to_mask
    t.umap_fn = lambda mu, cov, labels: np.zeros((size, size), dtype=np.float32)
    t.skew_umap_fn = lambda mu, cov, alpha, labels: (mu, np.zeros((size, size), dtype=np.float32))
    batch = {Tags.img: img.to(DEV), ContourTags.contour: contour.to(DEV), Tags.id: ["a", "b"]}
    out = t.predict(img.to(DEV))
    mu, cov = out[0], out[1]
    assert mu.shape == (n, 1, k, 2) and cov.shape == (n, 1, k, 2, 2)
    samples = t.sample(*out, t_a) if kind == "dsnt-al" else t.sample(out[0], out[1], out[2], t_a)
    assert samples.shape == (n, 1, t_a, k, 2) and np.isfinite(samples).all()
    # samples scatter around the predicted means (random-init network: wide distributions, so a loose bound)
    assert np.abs(samples.mean(axis=2) - mu.numpy()).max() < size
    res = t.predict_step(batch, 0)
    cs = res.contour_samples
    assert cs.shape[0] == n and cs.shape[-2:] == (k, 2) and np.isfinite(cs).all()
    assert res.mu.shape == (n, k, 2) and res.cov.shape == (n, k, 2, 2)
    assert res.post_mu.shape[-2:] == (k, 2) and np.isfinite(res.post_cov).all()
    assert set(res.point_uncertainty) >= {"cov_xx", "cov_yy", "cov_det", "cov_eigval_sum"}
    # a15: al / ep split and the sample covariances of THIS step's draws vs the loop-for-loop restatement of the
    # reference (oracle/predict_stats.py; aleatoric.py:88-108, aleatoric_skew.py:65-82).  predict() is deterministic at
    # t_e = 1 without dropout, so the step's (mu, cov[, alpha]) are `out`.
    from oracle import predict_stats as PS
    if kind == "dsnt-al":
        ref = PS.aleatoric_stats(out[0], out[1], cs)
    else:
        ref = PS.aleatoric_skew_stats(out[0], out[1], out[2], cs)
        assert np.allclose(res.alpha, ref["alpha"], rtol=1e-5, atol=1e-6)
    # two predict() calls differ by the rounding order of the f32 atomics in the InstanceNorm statistics (~1e-5 of a
    # covariance's scale), so the tolerances are relative to each tensor's largest entry
    def close(a, b, tol):
        return np.abs(np.asarray(a, dtype=np.float64) - b).max() <= tol * np.abs(b).max()
    assert close(res.mu, ref["mu"], 1e-4) and close(res.cov, ref["cov"], 2e-4)
    assert close(res.post_mu, ref["post_mu"], 1e-5) and close(res.post_cov, ref["post_cov"], 1e-5)


@pytest.mark.parametrize("kind", ["dsnt-al", "dsnt-skew"])
def test_predict_step_device_masks_and_entropy(golden_dir, kind):
    """with the CAMUS converter (the default) every sampled contour of the step is rasterised by
    cu_contour_masks and the entropy map reduced by cu_mask_entropy; both must equal the reference's per-contour
    USContourToMask + sample_entropy (oracle/masks.py) on the very contours the step sampled."""
    from oracle import masks as MO
    from contour_uncertainty._compat import ContourTags, Tags
    from contour_uncertainty.data.camus.utils import USContourToMask
    n, size, k, t_a = 2, 64, 21, 12
    img, contour = synthetic_batch(n, size, k, seed=5)
    torch.manual_seed(2)
    t = make_task(kind, 6, size, "f32").to(DEV).eval()
    t.hparams.psm_path = str(golden_dir / "camus-cont_psm_11_no_std.npz")
    t.hparams.t_a = t_a
    assert isinstance(t.contour_to_mask_fn, USContourToMask)
    t.contour_to_mask_fn = staticmethod(USContourToMask())          # how on_predict_start binds the datamodule's
    t.umap_fn = lambda mu, cov, labels: np.zeros((size, size), dtype=np.float32)
    t.skew_umap_fn = lambda mu, cov, alpha, labels: (mu, np.zeros((size, size), dtype=np.float32))
    res = t.predict_step({Tags.img: img.to(DEV), ContourTags.contour: contour.to(DEV), Tags.id: ["a", "b"]}, 0)
    cs = res.contour_samples
    t_a = 25 if kind == "dsnt-skew" else t_a          # hard-coded in the reference's skew predict step (aleatoric_skew.py:63)
    assert res.pred_samples.shape == (n, 1, t_a, size, size) and res.entropy_map.shape == (n, size, size)
    bad = 0
    for i in range(n):
        ref = np.stack([MO.us_contour_to_mask(cs[i, 0, j], (size, size)) for j in range(t_a)])
        bad += int((ref != res.pred_samples[i, 0]).sum())
        ent = MO.sample_entropy(res.pred_samples[i, 0][:, None].astype(float))
        assert np.allclose(res.entropy_map[i], ent, atol=1e-5)
    assert bad <= 2
    assert res.pred.shape == (n, size, size)
    # single-contour call keeps the reference signature and agrees with the batch
    one = USContourToMask()(cs[0, 0, 0], (size, size), None, apply_argmax=False)
    assert one.shape == (1, size, size) and (one[0] == res.pred_samples[0, 0, 0]).all()


def test_inference_forward_drops_layer_records():
    """Under no_grad every layer keeps only its activated output: same logits, and the peak memory of the
    forward is well below that of a forward that has to keep z + statistics + activations of every layer."""
    t = make_task("dsnt-al", 6, 128, "bf16").to(DEV).eval()
    x = torch.rand(16, 1, 128, 128, device=DEV)
    with torch.no_grad():
        t.model(x)                      # the persistent buffers (operand copies, scratch) exist before anything is measured
    peaks, outs = [], []
    for grad in (True, False):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        with torch.set_grad_enabled(grad):
            y = t.model(x)
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - base)
        outs.append(y.detach().float().clone())
        assert (y.grad_fn is not None) == grad
        del y
    # same kernels either way; the f32 atomics of the InstanceNorm statistics make two runs differ in their last bits
    assert float((outs[0] - outs[1]).abs().max()) <= 5e-2 * float(outs[0].abs().max())
    assert peaks[1] < 0.6 * peaks[0], peaks
