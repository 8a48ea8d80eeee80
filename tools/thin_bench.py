"""3x3 stride-1 convolutions of the two thin levels (256^2 x 32, 128^2 x 64): forward, concat forward, input gradient
with two destinations -- timing experiment.  CU_CONV_DMA_MINC=128 (tuning build) restores the register-staged kernel."""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3, TAPS3_D
DEV = "cuda"
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n = 64
for size, c in ((256, 32), (128, 64), (64, 128)):
    dt = torch.bfloat16
    x0 = ops.Act(torch.randn(n, size, size, c, device=DEV).to(dt), None, 1.0)
    x1 = ops.Act(torch.randn(n, size, size, c, device=DEV).to(dt), None, 1.0)
    z = torch.empty(n, size, size, c, device=DEV, dtype=dt)
    z2 = torch.empty(n, size, size, c, device=DEV, dtype=dt)
    for name, srcs, co, dsts, taps, acc in (("fwd c->c", [x0], c, [z], TAPS3, (0, 0)), ("fwd c+c->c", [x0, x1], c, [z], TAPS3, (0, 0)),
                                        ("dgrad c->c", [x0], c, [z], TAPS3_D, (0, 0)), ("dgrad c->c|c", [x0], 2 * c, [z, z2], TAPS3_D, (0, 0))):
        ci = sum(s.z.shape[3] for s in srcs)
        w = torch.randn(co, ci, 3, 3, device=DEV) / math.sqrt(9 * ci)
        wf, wd = ops.weight_prep(w, "conv", dt)
        us = bench(lambda: ops.conv_gemm(srcs, wf, None, grid=(size, size), in_stride=1, taps=taps, dsts=dsts,
                                         dst_cols=[d.shape[3] for d in dsts], accum=acc))
        gb = (n * size * size * (ci + co) * 2) / 1e9
        print(f"{size:4d}^2 C{c:<4d} {name:14s} {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s  {2.0*n*size*size*9*ci*co/us/1e6:7.1f} TFLOP/s", flush=True)
