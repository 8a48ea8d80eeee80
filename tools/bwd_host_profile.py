"""cProfile INSIDE the autograd worker thread: the host cost of _UNetFn.backward / _ConfidenceFn.backward (tools/cpu_bound.py only sees
the calling thread).  python tools/bwd_host_profile.py > profiles/r04_bwd_host_profile.txt"""
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
from contour_uncertainty.models.nnUnet import unet2

dev = torch.device("cuda", 0)
task, _ = build_task(256, "bf16", "dsnt-skew")
task = task.to(dev)
opt = task.configure_optimizers()["optimizer"]
img, contour = synthetic_batch(64, 256, 21, seed=1234)
b = {"img": img.to(dev), "contour": contour.to(dev)}
pr = cProfile.Profile()
on = [False]
spent = [0.0, 0]


def wrap(cls):
    orig = cls.backward

    def timed(ctx, *a):
        if not on[0]:
            return orig(ctx, *a)
        t0 = time.perf_counter()
        pr.enable()
        try:
            return orig(ctx, *a)
        finally:
            pr.disable()
            spent[0] += time.perf_counter() - t0
            spent[1] += 1
    cls.backward = staticmethod(timed)


wrap(unet2._UNetFn)
wrap(unet2._ConfidenceFn)


def step(i):
    opt.zero_grad(set_to_none=True)
    out = task.training_step(b, i)
    out["loss"].backward()
    opt.step()


for i in range(10):
    step(i)
torch.cuda.synchronize()
on[0] = True
K = 10
for i in range(K):
    step(i)
torch.cuda.synchronize()
print(f"# {1e3 * spent[0] / K:.2f} ms per step inside the two backward functions (with the profiler's own overhead), {K} steps")
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(40)
print("\n".join(s.getvalue().splitlines()[4:60]))
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_stats(30)
print("\n".join(s.getvalue().splitlines()[4:45]))
