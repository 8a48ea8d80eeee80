"""Throughput of the contour -> mask -> entropy kernels, with the scipy oracle timed beside it.
    python tools/masks_bench.py [--frames 16] [--samples 1024] [--json out.json]"""
import argparse, json, sys, time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu-masks", type=int, default=2000)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    from cu_hip import ops
    from oracle import masks as M
    g = np.random.default_rng(0)
    t = np.linspace(0, np.pi, 21)
    base = np.stack([128 + 60 * np.cos(t), 170 - 90 * np.sin(t)], -1)
    pts = base[None, None] + g.normal(size=(a.frames, a.samples, 21, 2)) * 3.0
    dev = torch.tensor(pts.reshape(-1, 21, 2), dtype=torch.float32).cuda()
    res = {}
    for name, fn in [("masks_packed", lambda: ops.contour_masks(dev, 256, 256, as_bytes=False)),
                     ("masks_packed_and_bytes", lambda: ops.contour_masks(dev, 256, 256))]:
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        res[name] = {"ms": ms, "masks_per_s": dev.shape[0] / ms * 1e3}
    packed = ops.contour_masks(dev, 256, 256, as_bytes=False)[0]
    ops.mask_entropy(packed, a.frames, 256); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.mask_entropy(packed, a.frames, 256)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    res["entropy"] = {"ms": ms, "GB_per_s": packed.numel() * 4 / ms / 1e6}
    n = min(a.cpu_masks, dev.shape[0])
    flat = pts.reshape(-1, 21, 2).astype(np.float32)
    t0 = time.time()
    for i in range(n):
        M.reconstruction(flat[i], 256, 256)
    dt = time.time() - t0
    res["cpu_oracle"] = {"masks": n, "s": dt, "masks_per_s": n / dt, "cores": 1}
    res["config"] = {"frames": a.frames, "samples": a.samples, "K": 21, "H": 256, "W": 256}
    line = json.dumps(res)
    print(line)
    if a.json:
        Path(a.json).parent.mkdir(parents=True, exist_ok=True)
        Path(a.json).write_text(line + "\n")


if __name__ == "__main__":
    main()
