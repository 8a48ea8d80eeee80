"""Potential of running two independent half-batch chains on two HIP streams (timing experiment): one task with batch 64
against two tasks with batch 32 each, enqueued alternately from one host thread."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from bench import build_task
from contour_uncertainty.data.synthetic import synthetic_batch
dev = torch.device("cuda", 0)

def make(batch, seed):
    task, _ = build_task(256, "bf16", "dsnt-skew")
    task = task.to(dev)
    opt = task.configure_optimizers()["optimizer"]
    img, contour = synthetic_batch(batch, 256, 21, seed=seed)
    return task, opt, {"img": img.to(dev), "contour": contour.to(dev)}

def step(t, i):
    task, opt, b = t
    opt.zero_grad(set_to_none=True)
    out = task.training_step(b, i)
    out["loss"].backward()
    opt.step()

def run(chains, streams, steps=20, warm=5):
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for c, s in zip(chains, streams):
            with torch.cuda.stream(s):
                step(c, i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

one = make(64, 1)
ms = run([one], [torch.cuda.current_stream()])
print(f"one chain, batch 64: {ms:.2f} ms/step  {64 / ms * 1e3:.0f} img/s", flush=True)
del one
torch.cuda.empty_cache()
a, b = make(32, 1), make(32, 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ms = run([a], [s1])
print(f"one chain, batch 32: {ms:.2f} ms/step  {32 / ms * 1e3:.0f} img/s", flush=True)
ms = run([a, b], [s1, s2])
print(f"two chains, batch 32 each: {ms:.2f} ms/step-pair  {64 / ms * 1e3:.0f} img/s", flush=True)
