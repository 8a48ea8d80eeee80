"""Size-independent properties at BASELINE.json's full configuration (dsnt-skew, 256x256x1, K=21, 8-stage unet2, batch 64,
bf16): the oracle cannot run this size in seconds, so the hot path is held to properties the domain offers.

  * InstanceNorm is per image and the loss is a mean over N*K points, so the network output of an image does not depend
    on its batch neighbours (batch-permutation equivariance), and the gradient of a batch is the mean of the gradients of
    its halves;
  * mu lies inside the image, Sigma is symmetric positive definite, the loss is finite and falls on a fixed batch.
Noise floors (f32 atomics in the InstanceNorm statistics and the weight gradients, amplified by a random-init bf16
network) are measured in the same test by repeating the identical computation."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
N, SIZE, K = 64, 256, 21


@pytest.fixture(scope="module")
def setup():
    from bench import build_task
    from oracle.step import synthetic_batch
    task, n_stages = build_task(SIZE, "bf16", "dsnt-skew")
    assert n_stages == 8
    task = task.to(DEV)
    img, contour = synthetic_batch(N, SIZE, K, seed=77)
    return task, img.to(DEV), contour.to(DEV)


def _flat_grads(task):
    return torch.cat([p.grad.detach().flatten().float() for p in task.parameters() if p.grad is not None])


def _step_grads(task, img, contour):
    task.zero_grad(set_to_none=True)
    out = task.training_step({"img": img, "contour": contour}, 0)
    out["loss"].backward()
    return float(out["loss"].detach()), _flat_grads(task).clone()


def test_outputs_are_valid_and_batch_permutation_equivariant(setup):
    task, img, _ = setup
    task.eval()
    mu, cov, alpha = task.predict(img)
    mu2, cov2, _ = task.predict(img)                       # run-to-run noise floor
    assert mu.shape == (N, 1, K, 2) and cov.shape == (N, 1, K, 2, 2) and alpha.shape == (N, 1, K, 2)
    assert torch.isfinite(mu).all() and torch.isfinite(cov).all() and torch.isfinite(alpha).all()
    assert (mu >= 0).all() and (mu <= SIZE - 1).all()
    assert torch.allclose(cov, cov.transpose(-1, -2))
    assert (cov[..., 0, 0] > 0).all() and (torch.linalg.det(cov) > 0).all()
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(3))
    mu_p, cov_p, _ = task.predict(img[perm.to(DEV)])
    noise = float((mu - mu2).abs().max())
    assert float((mu_p - mu[perm]).abs().max()) <= 5 * noise + 1e-3
    assert float((cov_p - cov[perm]).abs().max()) <= 5 * float((cov - cov2).abs().max()) + 1e-2 * float(cov.abs().max())
    task.train()


def test_gradient_of_the_batch_is_the_mean_of_its_halves(setup):
    task, img, contour = setup
    task.train()
    loss, g = _step_grads(task, img, contour)
    _, g_again = _step_grads(task, img, contour)           # noise floor of the identical computation
    la, ga = _step_grads(task, img[: N // 2], contour[: N // 2])
    lb, gb = _step_grads(task, img[N // 2:], contour[N // 2:])
    assert abs(loss - 0.5 * (la + lb)) <= 1e-3 * abs(loss)
    ref = 0.5 * (ga + gb)
    noise = float((g - g_again).norm() / g.norm())
    err = float((g - ref).norm() / g.norm())
    assert err <= 3 * noise + 2e-3, (err, noise)


def test_loss_falls_on_a_fixed_batch(setup):
    task, img, contour = setup
    task.train()
    opt = task.configure_optimizers()["optimizer"]
    losses = []
    for i in range(6):
        opt.zero_grad(set_to_none=True)
        out = task.training_step({"img": img, "contour": contour}, i)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"].detach()))
    assert all(map(lambda v: v == v and abs(v) < 1e6, losses))
    assert losses[-1] < losses[0]
