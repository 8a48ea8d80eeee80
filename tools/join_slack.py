"""Which stream ends the backward pass?  Events on the main, weight-gradient and reduction streams at the join of every backward
(engine.join_probe) and one at the start of the step: time from step start to each stream's last work, unprofiled.

    python tools/join_slack.py > profiles/r04_join_slack.txt
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from bench import build_task
from contour_uncertainty.data.synthetic import synthetic_batch

dev = torch.device("cuda", 0)
task, _ = build_task(256, "bf16", "dsnt-skew")
task = task.to(dev)
opt = task.configure_optimizers()["optimizer"]
img, contour = synthetic_batch(64, 256, 21, seed=1234)
b = {"img": img.to(dev), "contour": contour.to(dev)}
eng = task.model.engine
starts, fwd_end, ends = [], [], []


def step(i, probe):
    if probe:
        e = torch.cuda.Event(enable_timing=True); e.record(); starts.append(e)
    opt.zero_grad(set_to_none=True)
    out = task.training_step(b, i)
    if probe:
        e = torch.cuda.Event(enable_timing=True); e.record(); fwd_end.append(e)
    out["loss"].backward()
    opt.step()
    if probe:
        e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)


for i in range(20):
    step(i, False)
torch.cuda.synchronize()
eng.join_probe = []
for i in range(12):
    step(i, True)
torch.cuda.synchronize()
print("# per step, ms from the step's first launch: forward done | main stream's backward done | weight-gradient stream done | "
      "reduction stream done | step done (Adam)")
for k in range(2, 12):
    s = starts[k]
    m, sd, rd = eng.join_probe[k]
    f = lambda e: f"{s.elapsed_time(e):7.3f}" if e is not None else "   -   "     # noqa: E731
    print(f"step {k:2d}: {f(fwd_end[k])} | {f(m)} | {f(sd)} | {f(rd)} | {f(ends[k])}")
