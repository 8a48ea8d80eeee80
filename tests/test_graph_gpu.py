"""The training step as a hipGraph: replaying the captured step must do what the eager step does --
same losses step by step and the same weights after k steps (Adam's bias correction comes from the device counter)."""
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))


LR = 1e-6      # small enough that the seven steps see (nearly) the same gradient: at 1e-4 this random network goes through a
               # loss spike at step 2 (|g| x 300) whose size differs by several % from run to run, eager or not


def _build(capturable):
    """4 stages on 64 x 64 (8 x 8 at the bottom): a well-conditioned step.  The 6-stage test network normalises 2 x 2 maps,
    which amplifies the f32-atomics rounding noise until two EAGER runs of the same seven steps disagree by ~100 % in
    the Adam moments of the deep layers (tools/_graph_diag.py) -- nothing could be told apart there."""
    from test_model_gpu import make_task
    from oracle import unet as OU
    task = make_task("dsnt-al", 4, 64, "f32")
    task.hparams.optim = dict(task.hparams.optim, capturable=capturable, lr=LR)
    spec = OU.UNetSpec(strides=(1, 2, 2, 2))
    task.model.load_state_dict(OU.init_unet_state(spec, torch.Generator().manual_seed(0)), strict=True)
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
        assert abs(a - b) <= 1e-4 * abs(b), (got, losses)
    # seven Adam steps on a (nearly) constant gradient move every weight by about 7 * lr, in both runs, the same way
    da = eager.model.flat_params()[0] - w0
    db = task.model.flat_params()[0] - w0
    for d in (da, db):
        assert 5 * LR < float(d.abs().mean()) < 7.5 * LR, float(d.abs().mean())
    assert float((torch.sign(da) == torch.sign(db)).float().mean()) > 0.97

    def moments(o, t, key):
        return torch.cat([o.state[q][key].reshape(-1) for q in t.model.parameters() if q in o.state])

    for key, tol in (("exp_avg", 5e-3), ("exp_avg_sq", 5e-3)):
        ma, mb = moments(opt, eager, key), moments(copt, task, key)
        assert ma.shape == mb.shape and float((ma - mb).norm() / ma.norm()) < tol, (key, float((ma - mb).norm() / ma.norm()))
    st = copt.state[next(iter(task.model.parameters()))]
    assert float(st["step"]) == 7.0 and int(copt._steps_dev.item()) == 7


def test_changed_hyper_parameters_recapture_and_loaded_state_restarts_the_device_counter():
    """lr / betas / weight decay / grad_scale are scalar arguments of the captured launches -- a changed value
    must reach the replays (re-capture); ``load_state_dict`` on an optimizer that has stepped must not keep the old device
    step counter (the bias correction would be wrong)."""
    from contour_uncertainty.data.synthetic import synthetic_batch
    from cu_hip.graph import CapturedStep
    img, contour = synthetic_batch(2, 64, 21, seed=4)
    batch = {"img": img.cuda(), "contour": contour.cuda()}
    task = _build(True)
    copt = task.configure_optimizers()["optimizer"]
    step = CapturedStep(task, copt, batch, warmup=2)
    step.replay()
    w = task.model.flat_params()[0].clone()
    copt.param_groups[0]["lr"] = 0.0                 # scheduler / manual edit after the capture
    step.replay()
    assert step._hyper == copt.hyper_key() and copt.param_groups[0]["lr"] == 0.0
    # lr 0 and weight decay folded into the gradient only: the parameters must not move in the re-captured replay
    assert torch.equal(task.model.flat_params()[0], w)
    step.finish()
    sd = copt.state_dict()
    n_steps = float(next(iter(copt.state.values()))["step"])
    fresh = _build(True)
    fopt = fresh.configure_optimizers()["optimizer"]
    fresh.model.load_state_dict(task.model.state_dict())
    out = fresh.training_step(batch, 0); out["loss"].backward(); fopt.step()          # the optimizer has stepped once
    fopt.load_state_dict(sd)
    assert fopt._steps_dev is None and not fopt._flat
    fopt.zero_grad(set_to_none=True)
    out = fresh.training_step(batch, 1); out["loss"].backward(); fopt.step()
    assert int(fopt._steps_dev.item()) == int(n_steps) + 1
