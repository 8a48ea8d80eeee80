"""One wgrad launch (timing experiments)."""
import sys, math, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3_W
n, c, size = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 32
dt = torch.bfloat16
x = torch.randn(n, size, size, c, device="cuda").to(dt)
dz = torch.randn(n, size, size, c, device="cuda").to(dt)
dwk = torch.zeros(9, c, c, device="cuda")
for _ in range(3):
    ops.conv_wgrad([ops.Act(x, None, 1.0)], dz, dwk, grid=(size, size), in_stride=1, z_stride=1, taps=TAPS3_W, n_cols=c)
torch.cuda.synchronize()
