"""Diagnostics (not part of the product or the tests): per-layer gradient error of the HIP path vs the oracle."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from oracle import unet as OU
from oracle.step import OracleTask, synthetic_batch

sys.path.insert(0, str(ROOT / "tests"))


def grads_oracle(dtype, n=4, size=64, stages=6, seed=3):
    spec = OU.UNetSpec(strides=tuple([1] + [2] * (stages - 1)))
    img, contour = synthetic_batch(n, size, 21, seed=99)
    ot = OracleTask(spec, task="dsnt-skew", seed=seed)
    if dtype == torch.float64:
        ot.sd = {k: v.detach().double().requires_grad_(True) for k, v in ot.sd.items()}
        ot.skew_sd = {k: v.detach().double().requires_grad_(True) for k, v in ot.skew_sd.items()}
        img, contour = img.double(), contour.double()
    logs = ot.forward_loss(img, contour)
    logs["loss"].backward()
    return ot, {k: v.grad.detach().double() for k, v in ot.sd.items() if v.grad is not None}, float(logs["loss"])


def main():
    ot32, g32, l32 = grads_oracle(torch.float32)
    ot64, g64, l64 = grads_oracle(torch.float64)
    print("oracle loss f32/f64", l32, l64)
    res = {}
    if torch.cuda.is_available():
        from test_model_gpu import make_task
        img, contour = synthetic_batch(4, 64, 21, seed=99)
        for dt in ("f32", "bf16"):
            task = make_task("dsnt-skew", 6, 64, dt)
            task.model.load_state_dict({k: v.detach() for k, v in ot32.sd.items()}, strict=True)
            task.skew_block.load_state_dict({k: v.detach() for k, v in ot32.skew_sd.items()}, strict=True)
            task = task.cuda()
            out = task._shared_step({"img": img.cuda(), "contour": contour.cuda()}, 0)
            out["loss"].backward()
            res[dt] = ({k: p.grad.detach().double().cpu() for k, p in task.model.named_parameters() if p.grad is not None},
                       float(out["loss"]))
            print("hip", dt, "loss", res[dt][1])
    print(f"{'param':55s} {'f32ora':>9s} {'hipf32':>9s} {'hipbf16':>9s}   (relative L2 error vs f64 oracle)")
    for k in g64:
        if k.endswith("conv.bias") and "output" not in k:
            continue
        ref = g64[k]
        def e(x):
            return float((x - ref).norm() / ref.norm())
        row = [e(g32[k])]
        for dt in ("f32", "bf16"):
            row.append(e(res[dt][0][k]) if dt in res else float("nan"))
        print(f"{k:55s} " + " ".join(f"{v:9.2e}" for v in row))


if __name__ == "__main__":
    main()
