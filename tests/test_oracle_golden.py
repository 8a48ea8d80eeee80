"""Oracle (CPU restatement) vs golden vectors generated from the imported reference (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import head as H
from oracle import unet as U
from oracle.step import OracleTask, synthetic_batch


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def g_dsnt(golden_dir):
    return np.load(golden_dir / "dsnt_head.npz")


@pytest.mark.parametrize("size", [16, 64, 256])
def test_dsnt_fwd_bwd(g_dsnt, size):
    tag = f"s{size}"
    logits = T(g_dsnt[f"{tag}_logits"]).clone().requires_grad_(True)
    coords, var, covar = H.dsnt(H.flat_softmax(logits))
    assert torch.allclose(coords, T(g_dsnt[f"{tag}_coords"]), rtol=0, atol=1e-6)
    assert torch.allclose(var, T(g_dsnt[f"{tag}_var"]), rtol=1e-5, atol=1e-7)
    assert torch.allclose(covar, T(g_dsnt[f"{tag}_covar"]), rtol=1e-5, atol=1e-7)
    px = H.normalized_to_pixel_coordinates(coords, size)
    assert torch.allclose(px, T(g_dsnt[f"{tag}_pixel"]), rtol=0, atol=1e-4)
    loss = (coords * T(g_dsnt[f"{tag}_g_coords"])).sum() + (var * T(g_dsnt[f"{tag}_g_var"])).sum() \
        + (covar * T(g_dsnt[f"{tag}_g_covar"])).sum()
    loss.backward()
    if size <= 64:
        assert torch.allclose(logits.grad, T(g_dsnt[f"{tag}_dlogits"]), rtol=1e-4, atol=1e-8)
    else:
        assert torch.allclose(logits.grad[0, 0, 100], T(g_dsnt[f"{tag}_dlogits_row"]), rtol=1e-4, atol=1e-9)


def test_linspace_kat(g_dsnt):
    # reference docstring KAT, dsnt/utils.py:54-58
    assert np.allclose(g_dsnt["linspace4"], [-0.75, -0.25, 0.25, 0.75])
    assert torch.allclose(H.normalized_linspace(4), torch.tensor([-0.75, -0.25, 0.25, 0.75]))


@pytest.fixture(scope="module")
def g_nll(golden_dir):
    return np.load(golden_dir / "nll_heads.npz")


def test_skew_nll(g_nll):
    mu, y, cov, alpha = (T(g_nll[k]).clone() for k in ("mu", "y", "cov", "alpha"))
    mu.requires_grad_(True), cov.requires_grad_(True), alpha.requires_grad_(True)
    nll, t1, t2, t3 = H.skew_nll_terms(y, mu, cov, alpha)
    for got, key in ((nll, "skew_nll"), (t1, "skew_t1"), (t2, "skew_t2"), (t3, "skew_t3")):
        assert torch.allclose(got, T(g_nll[key]), rtol=1e-5, atol=1e-6), key
    nll.mean().backward()
    assert torch.allclose(mu.grad, T(g_nll["skew_dmu"]), rtol=1e-4, atol=1e-7)
    assert torch.allclose(alpha.grad, T(g_nll["skew_dalpha"]), rtol=1e-4, atol=1e-7)


def test_gauss_nll_literal_and_split(g_nll):
    mu, y, cov = (T(g_nll[k]) for k in ("mu", "y", "cov"))
    m = mu.shape[0]
    mu4, y4, cov4 = mu.view(1, m, 2), y.view(1, m, 2), cov.view(1, m, 2, 2)
    lit = H.gauss_nll(mu4, cov4, y4, literal_broadcast=True)
    split = H.gauss_nll(mu4, cov4, y4)
    ref = float(g_nll["gauss_loss"])
    assert abs(float(lit["loss"]) - ref) <= 1e-6 * abs(ref)
    # mean of the (NK,1,NK) broadcast == mean(t1) + mean(t2)
    assert abs(float(split["loss"]) - ref) <= 1e-5 * abs(ref)
    assert abs(float(split["loss_term1"]) - float(g_nll["gauss_t1_mean"])) < 1e-5
    assert abs(float(split["loss_term2"]) - float(g_nll["gauss_t2_mean"])) < 1e-4


def test_pdf_kat_vs_scipy(g_nll):
    # reference's manual check_scipy_equivalence constants (bivariatenormal.py:98-103)
    assert np.allclose(g_nll["kat_ref"], g_nll["kat_scipy"], rtol=1e-5)


@pytest.fixture(scope="module")
def g_small(golden_dir):
    return np.load(golden_dir / "unet_small.npz")


def small_net_state():
    spec = U.UNetSpec(in_channels=1, num_classes=5, strides=(1, 2, 2, 2))
    g = torch.Generator().manual_seed(11)
    sd = U.init_unet_state(spec, g)
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    x = torch.rand(2, 1, 32, 32, generator=g)
    return spec, sd, x


def test_unet_small_fwd_bwd(g_small):
    spec, sd, x = small_net_state()
    assert np.array_equal(x.numpy(), g_small["x"])
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits, bott = U.unet_forward(sd, x, spec, bottleneck_out=True)
    assert torch.allclose(logits, T(g_small["logits"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(bott, T(g_small["bottleneck"]), rtol=1e-4, atol=1e-5)
    ((logits * T(g_small["g_logits"])).sum() + (bott * T(g_small["g_bott"])).sum()).backward()
    names = [str(n) for n in g_small["grad_names"]]
    for i, n in enumerate(names):
        g = sd[n].grad
        ref = g_small["grad_stats"][i]
        l2 = float(g.double().pow(2).sum().sqrt())
        assert abs(l2 - ref[2]) <= 2e-4 * max(ref[2], 1e-6), n
        head = g.flatten()[:8].numpy()
        assert np.allclose(head, g_small["grad_head"][i][: len(head)], rtol=2e-3, atol=1e-5 * max(ref[2], 1e-6)), n
    # deep-supervision heads never receive gradients (SURVEY section 5 gotcha a)
    assert sorted(str(n) for n in g_small["no_grad_names"]) == sorted(
        n for n in sd if n.startswith("deep_supervision_heads"))


def test_unet_full_shapes_and_head(golden_dir):
    g = np.load(golden_dir / "unet_full.npz")
    spec = U.UNetSpec()
    shapes = U.param_shapes(spec)
    assert list(shapes.keys()) == [str(n) for n in g["param_names"]]
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params_unet"]) == 41298912
    assert sum(int(np.prod(s)) for s in U.confidence_param_shapes(42).values()) == int(g["n_params_skew"]) == 869802
    gen = torch.Generator().manual_seed(0)
    sd = U.init_unet_state(spec, gen)
    ssd = U.init_confidence_state(42, gen)
    x = torch.rand(1, 1, 256, 256, generator=gen)
    with torch.no_grad():
        logits, bott = U.unet_forward(sd, x, spec, bottleneck_out=True)
        a = U.confidence_forward(ssd, bott)
        coords, var, covar = H.dsnt(H.flat_softmax(logits))
    assert torch.allclose(bott, T(g["bottleneck"]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(a, T(g["alpha_raw"]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(logits[0, :, 128, :], T(g["logits_row"]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(coords, T(g["coords"]), rtol=0, atol=1e-5)
    assert torch.allclose(var, T(g["var"]), rtol=1e-4, atol=1e-7)
    assert torch.allclose(covar, T(g["covar"]), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("task", ["dsnt-skew", "dsnt-al"])
def test_train_step_matches_reference(golden_dir, task):
    g = np.load(golden_dir / "train_step.npz")
    spec = U.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    ot = OracleTask(spec, task=task, seed=0)
    img, contour = synthetic_batch(2, 64, 21, seed=1234)
    keys = ["loss", "distance_loss", "loss_term1", "loss_term2", "loss_term3", "alpha_norm"]
    for it in range(2):
        logs = ot.train_step(img, contour)
        ref = g[f"{task}_logs"][it]
        for j, r in enumerate(ref):
            assert abs(logs[keys[j]] - r) <= 2e-4 * max(1.0, abs(r)), (it, keys[j], logs[keys[j]], r)
    assert torch.allclose(ot.sd["output_block.conv.weight"].detach(), T(g[f"{task}_w_out"]), rtol=1e-3, atol=2e-5)
    assert torch.allclose(ot.sd["input_block.conv1.conv.weight"].detach(), T(g[f"{task}_w_in"]), rtol=1e-3, atol=2e-5)


def test_vital_unet_oracle_vs_reference_golden(golden_dir):
    """oracle/vital_unet.py against the fixture written from the reference module (vital/.../segmentation/unet.py):
    train-mode logits, every parameter gradient, the running statistics after the forward, eval-mode logits."""
    from oracle import vital_unet as OV
    g = np.load(golden_dir / "vital_unet.npz")
    gen = torch.Generator().manual_seed(23)
    sd = OV.init_state(1, 5, 32, gen)
    x = torch.from_numpy(g["x"])
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    state = dict(sd)
    state.update(params)
    logits = OV.forward(state, x, training=True)
    assert torch.allclose(logits, torch.from_numpy(g["logits"]), rtol=1e-4, atol=1e-5)
    (logits * torch.from_numpy(g["g_logits"])).sum().backward()
    for name, st, head in zip(g["grad_names"], g["grad_stats"], g["grad_head"]):
        gr = params[str(name)].grad
        l2 = float(gr.double().norm())
        assert abs(l2 - st[2]) <= 2e-4 * st[2] + 1e-7, name
        k = min(8, gr.numel())
        assert np.allclose(gr.flatten()[:k].numpy(), head[:k], rtol=2e-3, atol=1e-5 * max(st[2], 1e-3)), name
    for key in g.files:
        if key.startswith("rm:"):
            assert np.allclose(state[f"{key[3:]}.running_mean"].numpy(), g[key], rtol=1e-5, atol=1e-6)
        if key.startswith("rv:"):
            assert np.allclose(state[f"{key[3:]}.running_var"].numpy(), g[key], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        ev = OV.forward(state, x, training=False)
    assert torch.allclose(ev, torch.from_numpy(g["logits_eval"]), rtol=1e-4, atol=1e-5)
