"""Where the time of the wide / thin 3x3 stride-1 convolutions goes: the same launch with parts switched off
(tuning build only: CU_CONV_DBG bits 1 no stores, 2 no MFMA, 4 weights staged for the first chunk only, 8 halo likewise)."""
import os, sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
os.environ.setdefault("CONTOUR_HIP_LIB", str(ROOT / "contouring-uncertainty_amd" / "libcontour_hip_tuning.so"))
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3
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
MODES = [(0, "full"), (1, "no stores"), (2, "no MFMA"), (12, "no DMA after chunk 0"), (14, "no DMA, no MFMA"), (15, "nothing")]
if len(sys.argv) > 1:
    os.environ["CU_CONV_DNB"] = sys.argv[1]
    MODES = MODES[:1]
print("size C  " + "".join(f"{m[1]:>22s}" for m in MODES))
for size, c in ((256, 32), (128, 64), (64, 128), (32, 256), (16, 480)):
    dt = torch.bfloat16
    x0 = ops.Act(torch.randn(n, size, size, c, device=DEV).to(dt), None, 1.0)
    z = torch.empty(n, size, size, c, device=DEV, dtype=dt)
    w = torch.randn(c, c, 3, 3, device=DEV) / math.sqrt(9 * c)
    wf, wd = ops.weight_prep(w, "conv", dt)
    row = f"{size:4d} {c:<4d}"
    for bits, _ in MODES:
        os.environ["CU_CONV_DBG"] = str(bits)
        us = bench(lambda: ops.conv_gemm([x0], wf, None, grid=(size, size), in_stride=1, taps=TAPS3, dsts=[z],
                                         dst_cols=[c], accum=(0, 0)))
        row += f"{us:22.1f}"
    print(row, flush=True)
