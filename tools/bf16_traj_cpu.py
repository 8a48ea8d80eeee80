"""CPU side of VERDICT r3 item 1b: the SAME 20 Adam steps as tests/test_bf16_training_gpu.py / tools/bf16_spread.py (6-stage
net, 64x64, batch 2, dsnt-skew, fixed batch, seed-0 weights) on the CPU oracle, in float32 and with the bf16 STORAGE simulation
(oracle.unet._RoundBf16: conv outputs, activations, their gradients and the conv weights rounded to bf16, everything else f32).
No kernel of this repository runs here: if the simulated bf16 trajectory ends as far from the f32 one as the device's does, the
gap is a property of bf16 storage on this (chaotic, randomly initialised) problem, not of the HIP path.

A third arm perturbs the f32 run by ONE ulp-sized relative change of the input image (x * (1 + 1e-7)): the spread f32 itself shows
under a perturbation far below bf16's.

    python tools/bf16_traj_cpu.py > profiles/r04_bf16_traj_cpu.txt
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import unet as OU                                   # noqa: E402
from oracle.step import OracleTask, synthetic_batch              # noqa: E402

STEPS = 20


def run(round_bf16, eps=0.0, seed_w=0):
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    ot = OracleTask(spec, task="dsnt-skew", seed=seed_w)
    img, contour = synthetic_batch(2, 64, 21, seed=1234)
    img = img * (1.0 + eps)
    losses = []
    for _ in range(STEPS):
        ot.opt.zero_grad(set_to_none=True)
        logs = ot.forward_loss(img, contour, round_bf16=round_bf16)
        logs["loss"].backward()
        ot.opt.step()
        losses.append(float(logs["loss"].detach()))
    with torch.no_grad():
        losses.append(float(ot.forward_loss(img, contour, round_bf16=round_bf16)["loss"]))
    return losses


if __name__ == "__main__":
    torch.set_num_threads(8)
    f32 = run(False)
    print("f32 oracle            ", " ".join(f"{v:.3f}" for v in f32))
    for eps in (1e-7, -1e-7, 3e-7, 1e-6):
        p = run(False, eps)
        print(f"f32, image * (1{eps:+.0e})", f"final {p[-1]:.4f}   (vs f32 {p[-1] - f32[-1]:+.4f})", flush=True)
    sim = run(True)
    print("bf16 storage simulated", " ".join(f"{v:.3f}" for v in sim))
    print(f"final: f32 {f32[-1]:.4f}  bf16-sim {sim[-1]:.4f}  (bf16 - f32 = {sim[-1] - f32[-1]:+.4f})")
    for eps in (1e-7, -1e-7, 1e-6):
        p = run(True, eps)
        print(f"bf16-sim, image * (1{eps:+.0e})", f"final {p[-1]:.4f}   (vs f32 {p[-1] - f32[-1]:+.4f})", flush=True)
