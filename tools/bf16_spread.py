"""what is the run-to-run spread of the bf16 training trajectory made of?

20 identical Adam steps on one fixed batch (6-stage net, 64x64, batch 2, dsnt-skew: tests/test_bf16_training_gpu.py), final
contour NLL, RUNS times per configuration:
  * deterministic mode, f32 and bf16 (bit-reproducible: one number each);
  * default mode f32;
  * default mode bf16, and the same with ONE build switch turned off at a time: fused head, z-free first layer, skew head on
    its side stream, weight gradients on the second stream, small-map norm fusion, epilogue statistics / norm-backward sums.
If one switch owned the spread, turning it off would collapse it.

    python tools/bf16_spread.py [RUNS] > profiles/r04_bf16_spread.txt
"""
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "contouring-uncertainty_amd")):
    sys.path.insert(0, p)
DEV = "cuda"
STEPS = 20


def run(dtype, deterministic=False, off=()):
    from bench import build_task
    from contour_uncertainty.data.synthetic import synthetic_batch
    from contour_uncertainty.data.synthetic.weights import seeded_confidence_state, seeded_unet_state
    task, _ = build_task(64, dtype, "dsnt-skew")
    gen = torch.Generator().manual_seed(0)
    task.model.load_state_dict(seeded_unet_state(task.model, gen), strict=True)
    task.skew_block.load_state_dict(seeded_confidence_state(task.skew_block, gen), strict=True)
    task = task.to(DEV)
    eng = task.model.engine
    if deterministic:
        eng.deterministic = True
        task.skew_block.engine.deterministic = True
    for name in off:
        if name == "skew_side":
            task.skew_block.side_enabled = False
        else:
            assert hasattr(eng, name), name
            setattr(eng, name, False)
    img, contour = synthetic_batch(2, 64, 21, seed=1234)
    batch = {"img": img.to(DEV), "contour": contour.to(DEV)}
    opt = task.configure_optimizers()["optimizer"]
    losses = []
    for i in range(STEPS):
        opt.zero_grad(set_to_none=True)
        out = task.training_step(batch, i)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    with torch.no_grad():
        losses.append(float(task._shared_step(batch, 0)["loss"]))
    return losses


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    print(f"final contour NLL after {STEPS} Adam steps (start 6.87), {runs} runs per row: min / median / max / (max - min)")
    d32, d16 = run("f32", True)[-1], run("bf16", True)[-1]
    print(f"{'deterministic f32':44s} {d32:.4f}")
    print(f"{'deterministic bf16':44s} {d16:.4f}   (bf16 - f32 = {d16 - d32:+.4f})")
    rows = [("default f32", "f32", ()), ("default bf16", "bf16", ()), ("bf16, fused head off", "bf16", ("fused_head",)),
            ("bf16, z-free first layer off", "bf16", ("first_no_z",)), ("bf16, fused first layer off", "bf16", ("first_fused",)),
            ("bf16, skew head side stream off", "bf16", ("skew_side",)),
            ("bf16, second weight-gradient stream off", "bf16", ("side_wgrad",)),
            ("bf16, small-map norm fusion off", "bf16", ("small_norm",)),
            ("bf16, epilogue statistics off", "bf16", ("fused_stats",)),
            ("bf16, epilogue norm-backward sums off", "bf16", ("fused_norm_bwd",)),
            ("bf16, all of the above off", "bf16", ("fused_head", "first_no_z", "first_fused", "skew_side", "side_wgrad", "small_norm",
                                                    "fused_stats", "fused_norm_bwd"))]
    for label, dtype, off in rows:
        v = sorted(run(dtype, False, off)[-1] for _ in range(runs))
        print(f"{label:44s} {v[0]:.4f} / {statistics.median(v):.4f} / {v[-1]:.4f} / {v[-1] - v[0]:.4f}", flush=True)


if __name__ == "__main__":
    main()
