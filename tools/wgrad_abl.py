"""Where the time of the 3x3 stride-1 weight gradients goes: the same launch with parts switched off (tuning build only:
CU_CONV_DBG bits 1 no atomics, 2 no MFMA, 8 no staging after the first tile, 16 phase stamps of one workgroup)."""
import os, sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
os.environ.setdefault("CONTOUR_HIP_LIB", str(ROOT / "contouring-uncertainty_amd" / "libcontour_hip_tuning.so"))
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3_W
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
MODES = [(0, "full"), (1, "no atomics"), (2, "no MFMA"), (8, "no staging after tile 0"), (10, "neither"), (11, "nothing")]
print("size C  " + "".join(f"{m[1]:>24s}" for m in MODES))
shapes = ((256, 32), (128, 64), (64, 128), (32, 256), (16, 480))
for size, c in shapes:
    dt = torch.bfloat16
    x = torch.randn(n, size, size, c, device=DEV).to(dt)
    dz = torch.randn(n, size, size, c, device=DEV).to(dt)
    dwk = torch.zeros(9, c, c, device=DEV)
    row = f"{size:4d} {c:<4d}"
    for bits, _ in MODES:
        os.environ["CU_CONV_DBG"] = str(bits)
        us = bench(lambda: ops.conv_wgrad([ops.Act(x, None, 1.0)], dz, dwk, grid=(size, size), in_stride=1, z_stride=1,
                                          taps=TAPS3_W, n_cols=c))
        row += f"{us:24.1f}"
    print(row, flush=True)
os.environ["CU_CONV_DBG"] = "0"
print("== 128 x 64 blocks (CU_WGRAD_N128=1) vs default, with the result compared")
for size, c in ((64, 128), (32, 256), (16, 480)):
    x = torch.randn(n, size, size, c, device=DEV).to(torch.bfloat16)
    dz = torch.randn(n, size, size, c, device=DEV).to(torch.bfloat16)
    outs = []
    row = f"{size:4d} {c:<4d}"
    for v in ("0", "1"):
        os.environ["CU_WGRAD_N128"] = v
        dwk = torch.zeros(9, c, c, device=DEV)
        ops.conv_wgrad([ops.Act(x, None, 1.0)], dz, dwk, grid=(size, size), in_stride=1, z_stride=1, taps=TAPS3_W, n_cols=c)
        outs.append(dwk.clone())
        us = bench(lambda: ops.conv_wgrad([ops.Act(x, None, 1.0)], dz, dwk, grid=(size, size), in_stride=1, z_stride=1,
                                          taps=TAPS3_W, n_cols=c))
        row += f"   N128={v}: {us:7.1f} us"
    row += f"   rel diff {float((outs[0] - outs[1]).norm() / outs[0].norm()):.2e}"
    print(row, flush=True)
os.environ["CU_WGRAD_N128"] = "0"
os.environ["CU_CONV_DBG"] = "16"
sys.exit(0)
for size, c in shapes:
    x = torch.randn(n, size, size, c, device=DEV).to(torch.bfloat16)
    dz = torch.randn(n, size, size, c, device=DEV).to(torch.bfloat16)
    dwk = torch.zeros(9, c, c, device=DEV)
    print(f"-- stamps {size} C{c}", flush=True)
    ops.conv_wgrad([ops.Act(x, None, 1.0)], dz, dwk, grid=(size, size), in_stride=1, z_stride=1, taps=TAPS3_W, n_cols=c)
    torch.cuda.synchronize()
