"""the HEADLINE shape -- the real 8-stage unet2.yaml network at 256x256 -- through one whole training
step (forward, DSNT head, NLL, hand-written backward) against the CPU oracle (oracle/step.py, pinned to the reference by
tests/golden/*.npz) on the same seeded inputs.  The 256^2 / 128^2 levels select kernels no smaller network reaches (the
streaming thin-layer kernel, the 256 x 256 few-tap tile, the ring kernels, the resident / two-pass norm paths), so this is
the test of their COMPOSITION: stream joins, slab reuse, epilogue-statistics hand-over, two-pointer concat gradients.

  * f32 parity mode: every logged term to 3e-4 (the bound of the 64^2 reference-golden steps), every parameter gradient of
    the U-Net AND of the skew head (so the bottleneck gradient path is covered) in relative L2;
  * bf16 production mode (fused head, z-free first layer, side-stream skew head, small-map norm fusions -- the paths only
    bf16 takes): loss to 1e-3 of the f32 oracle; gradients against the f32 oracle AND against a CPU simulation that rounds
    the same tensors to bf16 (oracle.unet._RoundBf16).

LeakyReLU' is discontinuous at 0, so both oracles' backward passes are handed the device's own sign pattern
(oracle.unet._LeakyGivenMask) and the number of decisions that differ from the oracle's own is bounded
separately.  With the kink decisions shared, bf16 gradient errors are those of bf16 storage alone -- percent level, not the
tens of percent a flipped decision at the 2x2 / 4x4 levels causes."""
import pytest
import torch

from oracle import unet as OU
from oracle.step import OracleTask, synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SIZE, N = 256, 2


def _task(kind, dtype):
    from contour_uncertainty._compat import DataParameters
    from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
    from contour_uncertainty.task.regression.dsnt.dsnt_skew import DSNTSkew
    cfg = {"_target_": "contour_uncertainty.models.nnUnet.unet2.UNet", "kernels": [[3, 3]] * 8,
           "strides": [[1, 1]] + [[2, 2]] * 7, "patch_size": [256, 256], "drop_block": False, "deep_supervision": False,
           "compute_dtype": dtype}          # config/task/model/unet2.yaml
    cls = DSNTSkew if kind == "dsnt-skew" else DSNTAleatoric
    return cls(model=cfg, optim={"_target_": "torch.optim.Adam", "lr": 1e-3, "weight_decay": 1e-3}, choices={},
               data_params=DataParameters((1, SIZE, SIZE), (21, 2), [0, 1]), psm_path="unused.npy",
               seq_psm_path="unused.npy", t_a=25, t_e=1, covar=True)


def _masks(model):
    """LeakyReLU sign pattern of the device's last forward, per conv layer, from whatever the layer kept: its activation, or
    (fused head: the last layer keeps only z + statistics) sign(z * scale + shift)"""
    out = {}
    for prefix, rec in model.engine._last_ctx.convs.items():
        act = rec.out
        if act.a is not None:
            m = act.a.float() > 0
        else:
            m = (act.z.float() * act.stats[2][:, None, None, :] + act.stats[3][:, None, None, :]) > 0
        out[prefix] = m.permute(0, 3, 1, 2).cpu()
    return out


def _setup(kind, dtype):
    spec = OU.UNetSpec()
    assert spec.n_stages == 8
    ot = OracleTask(spec, task=kind, seed=0)
    task = _task(kind, dtype)
    task.model.load_state_dict({k: v.detach() for k, v in ot.sd.items()}, strict=True)
    if kind == "dsnt-skew":
        task.skew_block.load_state_dict({k: v.detach() for k, v in ot.skew_sd.items()}, strict=True)
    task = task.to(DEV)
    task.model.engine.keep_ctx = True          # keeps the layer records; does not change which kernels run
    img, contour = synthetic_batch(N, SIZE, 21, seed=1234)
    return spec, ot, task, img, contour


def _device_step(task, img, contour, warm=False):
    batch = {"img": img.to(DEV), "contour": contour.to(DEV)}
    if warm:      # the first call re-homes the skew head's parameters and therefore stays off its side stream: take the second
        task.training_step(batch, 0)["loss"].backward()
        task.zero_grad(set_to_none=True)
    out = task.training_step(batch, 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    grads = {f"model.{n}": p.grad.cpu() for n, p in task.model.named_parameters() if p.grad is not None}
    if hasattr(task, "skew_block"):
        grads.update({f"skew_block.{n}": p.grad.cpu() for n, p in task.skew_block.named_parameters() if p.grad is not None})
    return {k: float(v.detach()) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}, grads


def _oracle_grads(ot):
    g = {f"model.{n}": v.grad for n, v in ot.sd.items() if v.grad is not None}
    g.update({f"skew_block.{n}": v.grad for n, v in ot.skew_sd.items() if v.grad is not None})
    return g


def _flips(ot, spec, img, masks):
    taps = {}
    with torch.no_grad():
        OU.unet_forward({k: v.detach() for k, v in ot.sd.items()}, img, spec, taps=taps)
    return sum(int(((taps[f"{p}:a"] > 0) != m).sum()) for p, m in masks.items()), sum(m.numel() for m in masks.values())


def _bias_in_front_of_norm(name):
    return name.endswith("conv.bias") and "output_block" not in name


@pytest.mark.parametrize("kind", ["dsnt-skew", "dsnt-al"])
def test_full_size_training_step_f32_vs_oracle(kind):
    spec, ot, task, img, contour = _setup(kind, "f32")
    logs, grads = _device_step(task, img, contour)
    masks = _masks(task.model)
    ref = ot.forward_loss(img, contour, masks=masks)
    ref["loss"].backward()
    for k, r in ref.items():
        v, r = logs[f"train/{k}"], float(r)
        assert abs(v - r) <= 3e-4 * max(1.0, abs(r)), (k, v, r)
    n_flip, n_act = _flips(ot, spec, img, masks)
    assert n_flip <= 5e-4 * n_act, f"{n_flip} LeakyReLU decisions differ out of {n_act}"
    og = _oracle_grads(ot)
    assert set(og) == set(grads)
    worst = ("", 0.0)
    for name, b in og.items():
        if _bias_in_front_of_norm(name):
            assert float(grads[name].abs().max()) == 0.0        # analytically zero; ours is exact
            continue
        err = float((grads[name] - b).norm() / b.norm())
        worst = max(worst, (name, err), key=lambda t: t[1])
        # 1e-3.  Measured on MI355X: see profiles/r04_full_size_step_err.txt
        assert err <= 1e-3, (name, err)
    print(f"[full-size f32 {kind}] worst parameter-gradient error {worst[1]:.2e} at {worst[0]}; {n_flip} / {n_act} kink flips")


@pytest.mark.parametrize("kind", ["dsnt-skew", "dsnt-al"])
def test_full_size_training_step_bf16_vs_oracle(kind):
    spec, ot, task, img, contour = _setup(kind, "bf16")
    eng = task.model.engine
    logs, grads = _device_step(task, img, contour, warm=True)
    # the production paths really ran: fused head (no logits), z-free first layer, skew head on its own stream
    ctx = eng._last_ctx
    assert ctx.head is not None and ctx.convs["input_block.conv1"].no_z
    if kind == "dsnt-skew":
        assert task.skew_block._side is not None
    masks = _masks(task.model)
    ref = ot.forward_loss(img, contour, masks=masks)
    ref["loss"].backward()
    truth = _oracle_grads(ot)
    assert abs(logs["train/loss"] - float(ref["loss"])) <= 1e-3 * abs(float(ref["loss"]))
    for k, r in ref.items():
        assert abs(logs[f"train/{k}"] - float(r)) <= 5e-3 * max(1.0, abs(float(r))), (k, logs[f"train/{k}"], float(r))
    ot2 = OracleTask(spec, task=kind, seed=0)
    sim = ot2.forward_loss(img, contour, masks=masks, round_bf16=True)
    sim["loss"].backward()
    simg = _oracle_grads(ot2)
    worst = ("", 0.0, 0.0)
    num_h = num_s = den = 0.0
    tight = loose = 0
    for name, b in truth.items():
        if _bias_in_front_of_norm(name):
            continue
        e_hip = float((grads[name] - b).norm() / b.norm())
        e_sim = float((simg[name] - b).norm() / b.norm())
        num_h += float((grads[name] - b).norm()) ** 2
        num_s += float((simg[name] - b).norm()) ** 2
        den += float(b.norm()) ** 2
        if e_hip > worst[1]:
            worst = (name, e_hip, e_sim)
        if e_sim < 0.25:
            # with the kink decisions shared, what is left is bf16 storage of z, a and their gradients: the device may not be
            # worse than 1.5 x the CPU simulation of exactly that, + 1 % for the tensors the simulation keeps in f32 and the
            # device does not (bf16 operand copies of dz in the weight gradients, the fused head's bf16 g)
            tight += 1
            assert e_hip <= 1.5 * e_sim + 1e-2, (name, e_hip, e_sim)
        else:
            # tensors whose gradient bf16 storage ALONE moves by 25 % and more (the early, wide layers behind seven levels of
            # 2x2 ... 16x16 InstanceNorms, and their norm parameters): two bf16 realisations differ from each other as much as
            # from the truth (measured: 1.06 vs 0.58 on one run, 0.64 vs 0.92 on another), so only the order of magnitude is held
            loose += 1
            assert e_hip <= 2.5 * e_sim + 5e-2, (name, e_hip, e_sim)
    e_hip_all, e_sim_all = (num_h / den) ** 0.5, (num_s / den) ** 0.5
    assert tight >= 20, (tight, loose)                       # the tight bound really covers a substantial part of the network
    assert e_hip_all <= 1.5 * e_sim_all + 1e-2, (e_hip_all, e_sim_all)         # all parameters as one vector
    print(f"[full-size bf16 {kind}] worst parameter-gradient error {worst[1]:.3f} (simulation {worst[2]:.3f}) at {worst[0]}; "
          f"all parameters as one vector {e_hip_all:.3f} (simulation {e_sim_all:.3f}); {tight} tensors held to 1.5x + 1 %, {loose} to 2.5x + 5 %")
