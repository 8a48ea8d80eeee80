"""The training step as a hipGraph: replaying the captured step must do what the eager step does --
same losses step by step and the same weights after k steps (Adam's bias correction comes from the device counter)."""
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))


LR = 5e-5      # small enough that two runs of the same sequence stay together (at the config's 1e-3 the first Adam steps of a
               # random network amplify the f32-atomics rounding noise to 8 % of the loss: tools/_graph_diag.py)


def _build(capturable):
    from test_model_gpu import make_task
    from oracle import unet as OU
    task = make_task("dsnt-skew", 6, 64, "f32")
    task.hparams.optim = dict(task.hparams.optim, capturable=capturable, lr=LR)
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    gen = torch.Generator().manual_seed(0)
    task.model.load_state_dict(OU.init_unet_state(spec, gen), strict=True)
    task.skew_block.load_state_dict(OU.init_confidence_state(42, gen), strict=True)
    return task.to("cuda")


def test_device_step_counter_gives_the_same_update_as_the_host_counter():
    """cu_adam_step_dev (bias correction from the device counter) == cu_adam_step (from the host argument) up to the
    last bits of powf (libm on the host, ocml on the device)"""
    from cu_hip.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(3, 4099, generator=g)
    grads = [torch.randn(3, 4099, generator=g) for _ in range(6)]
    out = []
    for capturable in (False, True):
        w = torch.nn.Parameter(w0.clone().cuda())
        opt = FusedAdam([w], lr=1e-3, weight_decay=1e-3, capturable=capturable)
        for gr in grads:
            w.grad = gr.cuda()
            opt.step()
        out.append((w.detach().clone(), opt.state[w]["exp_avg"].clone(), opt.state[w]["exp_avg_sq"].clone()))
    for a, b in zip(*out):
        torch.testing.assert_close(a, b, rtol=3e-6, atol=1e-8)
    assert int(opt._steps_dev.item()) == 6


def test_captured_step_replays_like_the_eager_step():
    from contour_uncertainty.data.synthetic import synthetic_batch
    from cu_hip.graph import CapturedStep
    img, contour = synthetic_batch(4, 64, 21, seed=3)
    batch = {"img": img.cuda(), "contour": contour.cuda()}
    # eager reference: warm-up count of CapturedStep (3) + 4 more steps
    eager = _build(False)
    w0 = eager.model.flat_params()[0].clone()
    opt = eager.configure_optimizers()["optimizer"]
    losses = []
    for i in range(7):
        opt.zero_grad(set_to_none=True)
        out = eager.training_step(batch, i)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    task = _build(True)
    copt = task.configure_optimizers()["optimizer"]
    assert copt.capturable
    step = CapturedStep(task, copt, batch, warmup=3)
    got = []
    for _ in range(4):
        step.replay()
        got.append(float(step.logs["loss"]))
    step.finish()
    # the loss of replay k is the loss of eager step 3 + k
    for a, b in zip(got, losses[3:]):
        assert abs(a - b) <= 5e-3 * abs(b), (got, losses)
    assert got[-1] < got[0] < losses[0]                 # and it is training, not replaying one frozen step
    # Both runs moved the weights (about 7 * lr each) and accumulated the same Adam moments.  The moments are compared, not
    # the weights: Adam normalises every element to a step of about lr whatever its gradient, so elements whose gradient is
    # f32-atomics rounding noise take either sign from run to run (a third of the L2 norm of the weight change), while
    # the moments are dominated by the elements that carry signal
    da = eager.model.flat_params()[0] - w0
    db = task.model.flat_params()[0] - w0
    assert float(da.abs().mean()) > 2 * LR and float(db.abs().mean()) > 2 * LR

    def moments(o, t, key):
        return torch.cat([o.state[q][key].reshape(-1) for q in t.model.parameters() if q in o.state])

    for key, tol in (("exp_avg", 0.05), ("exp_avg_sq", 0.05)):
        ma, mb = moments(opt, eager, key), moments(copt, task, key)
        assert ma.shape == mb.shape and float((ma - mb).norm() / ma.norm()) < tol, (key, float((ma - mb).norm() / ma.norm()))
    st = copt.state[next(iter(task.model.parameters()))]
    assert float(st["step"]) == 7.0 and int(copt._steps_dev.item()) == 7
