"""8x8 / 16x16 layers of the bottom of the U on the 8-wave LDS-DMA kernel vs the generic kernel (tuning build:
CU_CONV_DMA_MINWG = fewest workgroups for which the DMA kernel is taken)."""
import os, sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
os.environ.setdefault("CONTOUR_HIP_LIB", str(ROOT / "contouring-uncertainty_amd" / "libcontour_hip_tuning.so"))
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3, TAPS3_D
DEV = "cuda"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n, dt = 64, torch.bfloat16
for size, ci, co, two in ((8, 480, 480, False), (8, 960, 480, False), (8, 480, 960, True), (16, 480, 480, False), (16, 480, 960, True)):
    if ci == 960:
        srcs = [ops.Act(torch.randn(n, size, size, 480, device=DEV).to(dt), None, 1.0) for _ in range(2)]
    else:
        srcs = [ops.Act(torch.randn(n, size, size, ci, device=DEV).to(dt), None, 1.0)]
    w = torch.randn(co, ci, 3, 3, device=DEV) / math.sqrt(9 * ci)
    wf, _ = ops.weight_prep(w, "conv", dt)
    dsts = [torch.empty(n, size, size, 480, device=DEV, dtype=dt) for _ in range(2 if two else 1)]
    row = f"{size:3d}x{size:<3d} C{ci}->{co}{' (2 dst)' if two else ''}"
    for minwg in (128, 64, 32, 16):
        os.environ["CU_CONV_DMA_MINWG"] = str(minwg)
        us = bench(lambda: ops.conv_gemm(srcs, wf, None, grid=(size, size), in_stride=1, taps=TAPS3, dsts=dsts,
                                         dst_cols=[480] * len(dsts)))
        row += f"   minwg {minwg:3d}: {us:6.1f} us"
    print(row, flush=True)
