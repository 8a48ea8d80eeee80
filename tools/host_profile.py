"""Where does the HOST spend its time enqueueing one eager training step?  cProfile over 10 steps (no synchronisation inside),
top functions by own time and by cumulative time.

    python tools/host_profile.py > profiles/r04_host_profile.txt
"""
import cProfile
import io
import pstats
import sys
import time
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


def step(i):
    opt.zero_grad(set_to_none=True)
    out = task.training_step(b, i)
    out["loss"].backward()
    opt.step()


for i in range(5):
    step(i)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for i in range(K):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"# enqueue {1e3 * (t1 - t0) / K:.2f} ms per step, with the final synchronisation {1e3 * (t2 - t0) / K:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    step(i)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(35)
    print(f"# ---- by {key} ({K} steps)")
    print("\n".join(s.getvalue().splitlines()[4:50]))
