"""Where do a training step's launches come from?  One eager bench step (dsnt-skew, 256x256, batch 64, bf16) under
torch.profiler: kernel launches per step by name, and for the torch-side kernels (fills, copies, elementwise glue) the Python
source line that issued them.

    python tools/launch_census.py [batch] > profiles/r04_launch_census.txt
"""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "contouring-uncertainty_amd")):
    sys.path.insert(0, p)


def main():
    from bench import build_task
    from contour_uncertainty.data.synthetic import synthetic_batch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    task, _ = build_task(256, "bf16", "dsnt-skew")
    task = task.cuda()
    img, contour = synthetic_batch(n, 256, 21, seed=1)
    batch = {"img": img.cuda(), "contour": contour.cuda()}
    opt = task.configure_optimizers()["optimizer"]

    def step(i):
        opt.zero_grad(set_to_none=True)
        out = task.training_step(batch, i)
        out["loss"].backward()
        opt.step()

    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    steps = 3
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
    kernels = collections.Counter()
    ktime = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            kernels[ev.name] += 1
            ktime[ev.name] += ev.device_time if hasattr(ev, "device_time") else ev.cuda_time
    total = sum(kernels.values())
    print(f"# batch {n}: {total / steps:.1f} device activities per step (kernels + memcpy/memset), {steps} steps profiled")
    print("# per step | us per step | name")
    for name, c in sorted(kernels.items(), key=lambda kv: -kv[1]):
        print(f"{c / steps:7.1f} {ktime[name] / steps:10.1f}  {name[:150]}")
    # torch-side ops that launch something: aten op + innermost repository frame
    print("\n# aten ops with device work, per step, by the innermost repository source line")
    by_src = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        if not getattr(ev, "kernels", []):
            continue
        if ev.cpu_children and any(c.name.startswith("aten::") and getattr(c, "kernels", []) for c in ev.cpu_children):
            continue
        frame = next((f for f in ev.stack if "/repo/" in f and "site-packages" not in f), ev.stack[0] if ev.stack else "?")
        by_src[(ev.name, frame.strip()[-110:])] += 1
    for (name, frame), c in sorted(by_src.items(), key=lambda kv: -kv[1]):
        print(f"{c / steps:6.1f}  {name:28s} {frame}")


if __name__ == "__main__":
    main()
