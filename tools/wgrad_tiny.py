"""Weight gradients of the tiny feature maps (2x2 ... 8x8, 480 channels, batch 64): 64 x 64 against 32 x 32 blocks of dW
(tuning build, CU_WGRAD_SMALLPX)."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
os.environ.setdefault("CONTOUR_HIP_LIB", str(ROOT / "contouring-uncertainty_amd" / "libcontour_hip_tuning.so"))
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3_W
DEV = "cuda"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n = 64
for size, ci, co in ((2, 480, 480), (4, 480, 480), (4, 960, 480), (8, 480, 480), (8, 960, 480), (16, 480, 480)):
    x = torch.randn(n, size, size, ci, device=DEV).to(torch.bfloat16)
    dz = torch.randn(n, size, size, co, device=DEV).to(torch.bfloat16)
    dwk = torch.zeros(9, co, ci, device=DEV)
    row = f"{size:3d}x{size:<3d} C{ci}->{co}"
    for small in (0, 1 << 30):
        os.environ["CU_WGRAD_SMALLPX"] = str(small)
        us = bench(lambda: ops.conv_wgrad([ops.Act(x, None, 1.0)], dz, dwk, grid=(size, size), in_stride=1, z_stride=1,
                                          taps=TAPS3_W, n_cols=co))
        row += f"   {'32x32' if small else '64x64'} {us:7.1f} us"
    print(row, flush=True)
