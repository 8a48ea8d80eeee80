"""Soak: N eager training steps of the bench configuration on fresh synthetic batches (8 different batches in rotation), loss and
gradient-norm sanity every 100 steps; fails on a non-finite value.  python tools/soak.py [steps] > profiles/r04_soak.txt"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from bench import build_task
from contour_uncertainty.data.synthetic import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)
task, _ = build_task(256, "bf16", "dsnt-skew")
task = task.to(dev)
opt = task.configure_optimizers()["optimizer"]
batches = []
for s in range(8):
    img, contour = synthetic_batch(64, 256, 21, seed=100 + s)
    batches.append({"img": img.to(dev), "contour": contour.to(dev)})
t0 = time.perf_counter()
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    out = task.training_step(batches[i % 8], i)
    out["loss"].backward()
    if i % 100 == 0 or i == steps - 1:
        gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in task.parameters() if p.grad is not None))
        loss = float(out["loss"].detach())
        print(f"step {i:5d}: loss {loss:9.4f}  |grad| {float(gn):10.4f}  "
              f"{time.perf_counter() - t0:7.1f} s", flush=True)
        assert loss == loss and abs(loss) < 1e6 and float(gn) == float(gn), "non-finite training state"
    opt.step()
torch.cuda.synchronize()
print(f"# {steps} steps in {time.perf_counter() - t0:.1f} s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
