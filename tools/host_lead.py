"""Is the host ahead of the GPU?  Marks (host clock, HIP event on the main stream) at the phase boundaries of the eager training
step, several steps without any synchronisation; lead = GPU time of the mark - host time of the mark (both from a common
synchronised origin): ~0 = the GPU reached the mark as soon as the host enqueued it (the GPU waits for the host there), large = the
host runs ahead.

    python tools/host_lead.py [batch] > profiles/r04_host_lead.txt
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from bench import build_task
from contour_uncertainty.data.synthetic import synthetic_batch
from contour_uncertainty.models.nnUnet import unet2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
task, _ = build_task(256, "bf16", "dsnt-skew")
task = task.to(dev)
opt = task.configure_optimizers()["optimizer"]
img, contour = synthetic_batch(B, 256, 21, seed=1234)
b = {"img": img.to(dev), "contour": contour.to(dev)}
marks = []
on = [False]


def mark(name):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, time.perf_counter(), e))


def wrap(cls, label):
    orig = cls.backward

    def timed(ctx, *a):
        mark(label + " in")
        out = orig(ctx, *a)
        mark(label + " out")
        return out
    cls.backward = staticmethod(timed)


wrap(unet2._UNetFn, "unet.backward")
wrap(unet2._ConfidenceFn, "confidence.backward")


def step(i):
    mark("step start")
    opt.zero_grad(set_to_none=True)
    out = task.training_step(b, i)
    mark("forward enqueued")
    out["loss"].backward()
    mark("backward enqueued")
    opt.step()
    mark("optimizer enqueued")


for i in range(8):
    step(i)
torch.cuda.synchronize()
on[0] = True
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 4
for i in range(K):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"# batch {B}: host enqueue {1e3 * (t1 - t0) / K:.2f} ms per step; with the final synchronisation {1e3 * (t2 - t0) / K:.2f} ms per step")
print("# mark | host ms | gpu ms | lead ms (gpu - host)")
for name, th, ev in marks:
    h = 1e3 * (th - t0)
    g = e0.elapsed_time(ev)
    print(f"{name:28s} {h:9.3f} {g:9.3f} {g - h:9.3f}")
