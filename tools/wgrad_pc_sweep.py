"""Producer / consumer split of the 3x3 weight-gradient kernel (tuning build): CU_WGRAD_PCE rounds issued by the computing
waves BEFORE their k-loop, CU_WGRAD_PCF percent of the remaining rounds issued by the producer waves (the rest by the
computing waves after the k-loop).  One process per setting (the knobs are read once).

    python tools/wgrad_pc_sweep.py > profiles/r03_wgrad_pc_sweep.txt
"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import os, sys
sys.path.insert(0, r"%(root)s"); sys.path.insert(0, r"%(root)s/contouring-uncertainty_amd")
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3_W
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = []
for size, c in ((64, 128), (32, 256), (16, 480)):
    x = torch.randn(64, size, size, c, device="cuda").to(torch.bfloat16)
    dz = torch.randn(64, size, size, c, device="cuda").to(torch.bfloat16)
    ws = torch.empty(24 << 20, device="cuda")
    out.append(bench(lambda: ops.conv_wgrad([ops.Act(x, None, 1.0)], dz, ws, grid=(size, size), in_stride=1, z_stride=1,
                                            taps=TAPS3_W, n_cols=c, parts=True)))
print(" ".join(f"{v:8.1f}" for v in out))
'''
print("# us per launch (partial-tile mode): 64^2 x 128, 32^2 x 256, 16^2 x 480 channels, batch 64")
for pce in (0, 4, 8):
    for pcf in (50, 70, 85, 100):
        env = dict(os.environ, CONTOUR_HIP_LIB=str(ROOT / "contouring-uncertainty_amd" / "libcontour_hip_tuning.so"),
                   CU_WGRAD_PCE=str(pce), CU_WGRAD_PCF=str(pcf))
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": str(ROOT)}], env=env, capture_output=True, text=True)
        print(f"PCE {pce:2d} PCF {pcf:3d}: {r.stdout.strip() or r.stderr[-200:]}", flush=True)
