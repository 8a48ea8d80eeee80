"""Is the step CPU-bound?  Enqueue time (no sync) vs GPU time of K steps."""
import sys, time
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
for i in range(3): step(i)
torch.cuda.synchronize()
K = 30
best = None
for rep in range(4):        # best of four: the host's share of a box varies
    t0 = time.perf_counter()
    for i in range(K): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cur = (1e3*(t1-t0)/K, 1e3*(t2-t0)/K)
    best = cur if best is None or cur[0] < best[0] else best
print(f"enqueue {best[0]:.2f} ms/step, total {best[1]:.2f} ms/step (best of 4 x {K} steps)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(3): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
