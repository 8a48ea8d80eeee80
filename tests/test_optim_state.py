"""FusedAdam keeps torch.optim.Adam's state layout: per-parameter step / exp_avg / exp_avg_sq that are views
into one flat buffer per contiguous run, rebuilt -- never silently zeroed -- after load_state_dict.

The CPU tests swap the cu_adam_step launch for a plain-torch statement of the same update (the kernel itself is held to
torch.optim.Adam by tests/test_kernels_gpu.py::test_adam_matches_torch); the ``gpu`` test runs the real kernel."""
import copy

import pytest
import torch


def _torch_adam_step(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale=1.0):
    gg = g * grad_scale + wd * p
    m.mul_(b1).add_(gg, alpha=1 - b1)
    v.mul_(b2).addcmul_(gg, gg, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.sub_(lr / bc1 * m / (v.sqrt() / bc2 ** 0.5 + eps))


def _flat_model(dev, seed=0):
    """three parameters living back to back in one flat buffer + one unused parameter (never gets a gradient)"""
    g = torch.Generator().manual_seed(seed)
    shapes = [(4, 3, 3, 3), (4,), (5, 4)]
    flat = torch.randn(sum(torch.Size(s).numel() for s in shapes), generator=g).to(dev)
    params, off = [], 0
    for s in shapes:
        n = torch.Size(s).numel()
        params.append(torch.nn.Parameter(flat[off:off + n].view(s)))
        off += n
    params.append(torch.nn.Parameter(torch.randn(7, generator=g).to(dev)))
    return flat, params


def _set_grads(params, seed, dev):
    g = torch.Generator().manual_seed(seed)
    n = sum(p.numel() for p in params[:3])
    gflat = torch.randn(n, generator=g).to(dev)
    off = 0
    for p in params[:3]:
        p.grad = gflat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    params[3].grad = None


def _run(dev, monkeypatch=None):
    from cu_hip import ops
    from cu_hip.optim import FusedAdam
    launches = []
    if monkeypatch is not None:
        def fake(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale=1.0):
            launches.append((p.numel(), step))
            _torch_adam_step(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale)
        monkeypatch.setattr(ops, "adam_step", fake)
    _, params = _flat_model(dev)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = FusedAdam(params, lr=1e-2, weight_decay=1e-3)
    ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-3)
    for it in range(3):
        _set_grads(params, 10 + it, dev)
        for p, r in zip(params, ref):
            r.grad = None if p.grad is None else p.grad.detach().clone()
        opt.step()
        ropt.step()
    for p, r in zip(params, ref):
        assert torch.allclose(p, r, rtol=1e-5, atol=1e-6)
    if monkeypatch is not None:
        assert launches == [(sum(p.numel() for p in params[:3]), s) for s in (1, 2, 3)]   # ONE launch per step
    # ---- the state is torch.optim.Adam's: per parameter, same keys, same values
    sd, rsd = opt.state_dict(), ropt.state_dict()
    assert sorted(sd["state"].keys()) == sorted(rsd["state"].keys()) == [0, 1, 2]          # the unused tensor has none
    for i in range(3):
        assert set(sd["state"][i].keys()) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(sd["state"][i]["step"]) == float(rsd["state"][i]["step"]) == 3.0
        assert sd["state"][i]["exp_avg"].shape == params[i].shape
        assert torch.allclose(sd["state"][i]["exp_avg"], rsd["state"][i]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(sd["state"][i]["exp_avg_sq"], rsd["state"][i]["exp_avg_sq"], rtol=1e-4, atol=1e-9)
    # ---- save / load round trip into a NEW optimizer over re-flattened parameters: step and moments survive
    saved = copy.deepcopy(sd)
    _, params2 = _flat_model(dev, seed=0)
    with torch.no_grad():
        for p2, p in zip(params2, params):
            p2.copy_(p)
    opt2 = FusedAdam(params2, lr=1e-2, weight_decay=1e-3)
    opt2.load_state_dict(saved)
    # ... and torch.optim.Adam's own state dict loads too (a reference checkpoint's optimizer_states entry)
    _, params3 = _flat_model(dev, seed=0)
    with torch.no_grad():
        for p3, p in zip(params3, params):
            p3.copy_(p)
    opt3 = FusedAdam(params3, lr=1e-2, weight_decay=1e-3)
    opt3.load_state_dict(copy.deepcopy(rsd))
    for it in range(3, 5):
        for ps, o in ((params, opt), (params2, opt2), (params3, opt3)):
            _set_grads(ps, 10 + it, dev)
            o.step()
        for p, r in zip(params, ref):
            r.grad = None if p.grad is None else p.grad.detach().clone()
        ropt.step()
    for p, p2, p3, r in zip(params, params2, params3, ref):
        assert torch.allclose(p2, p, rtol=1e-6, atol=1e-7), "resumed optimizer diverged: state was not carried over"
        assert torch.allclose(p3, r, rtol=1e-5, atol=1e-6)
        assert torch.allclose(p, r, rtol=1e-5, atol=1e-6)
    assert float(opt2.state[params2[0]]["step"]) == 5.0
    # the rebuilt moments are views of one flat buffer again (one launch per step)
    m0, m1 = opt2.state[params2[0]]["exp_avg"], opt2.state[params2[1]]["exp_avg"]
    assert m1.data_ptr() == m0.data_ptr() + 4 * m0.numel()
    # ---- a parameter that joins late (first gradient at step 6) gets its own bias correction
    for ps, o in ((params, opt),):
        _set_grads(ps, 99, dev)
        ps[3].grad = torch.ones_like(ps[3])
        o.step()
    for p, r in zip(params, ref):
        r.grad = p.grad.detach().clone()
    ropt.step()
    for p, r in zip(params, ref):
        assert torch.allclose(p, r, rtol=1e-5, atol=1e-6)
    assert float(opt.state[params[3]]["step"]) == 1.0 and float(opt.state[params[0]]["step"]) == 6.0


def test_state_layout_and_resume_cpu(monkeypatch):
    _run("cpu", monkeypatch)


@pytest.mark.gpu
def test_state_layout_and_resume_gpu():
    _run("cuda")


def test_step_plan_is_replayed_and_invalidated_cpu(monkeypatch):
    """Round 4: ``FusedAdam.step`` replays the launches of the previous step after checking that nothing moved (parameter list,
    addresses, gradient layout, which parameters have gradients).  Same updates as torch.optim.Adam through every transition:
    replay, gradients in separately allocated tensors (two launches instead of one), back to one flat buffer, a parameter whose
    gradient disappears, a new hyper-parameter."""
    from cu_hip import ops
    from cu_hip.optim import FusedAdam
    launches = []

    def fake(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale=1.0):
        launches.append((p.numel(), step))
        _torch_adam_step(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale)
    monkeypatch.setattr(ops, "adam_step", fake)
    _, params = _flat_model("cpu")
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = FusedAdam(params, lr=1e-2, weight_decay=1e-3)
    ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-3)
    full_paths = []
    orig_runs = FusedAdam._runs
    monkeypatch.setattr(FusedAdam, "_runs", staticmethod(lambda ps: (full_paths.append(1), orig_runs(ps))[1]))
    n3 = sum(p.numel() for p in params[:3])

    def both(it, mutate=None):
        _set_grads(params, 30 + it, "cpu")
        if mutate is not None:
            mutate()
        for p, r in zip(params, ref):
            r.grad = None if p.grad is None else p.grad.detach().clone()
        opt.step()
        ropt.step()
        for p, r in zip(params, ref):
            assert torch.allclose(p, r, rtol=1e-5, atol=1e-6), it

    both(0); both(1); both(2)
    assert len(full_paths) == 1 and launches == [(n3, 1), (n3, 2), (n3, 3)]            # steps 2 and 3 were replays

    def scatter():       # every gradient in a tensor of its own: the run structure changes
        for p in params[:3]:
            p.grad = p.grad.detach().clone()
    both(3, scatter)
    assert len(full_paths) == 2 and len(launches) > 4 and launches[-1][1] == 4
    both(4)          # one flat gradient buffer again: the three-launch plan still describes it (each launch's layout holds)
    assert len(full_paths) == 2 and [l[1] for l in launches[-3:]] == [5, 5, 5]
    opt._plans.clear()                                                                 # (a fresh plan merges them again)
    both(5)
    assert len(full_paths) == 3 and launches[-1] == (n3, 6)

    def drop():          # a parameter without a gradient this step (frozen layer): it must not move
        params[2].grad = None
    before = params[2].detach().clone()
    both(6, drop)
    assert len(full_paths) == 4 and torch.equal(params[2], before)
    opt.param_groups[0]["lr"] = ropt.param_groups[0]["lr"] = 3e-3                      # hyper-parameters are read every step
    both(7); both(8)
    assert float(opt.state[params[0]]["step"]) == 9.0 and float(opt.state[params[2]]["step"]) == 8.0
