"""True device durations (rocprofv3 kernel trace, not host-paired events) of the small-map InstanceNorm backward kernels:
run under   rocprofv3 --kernel-trace --stats -d gpurun_out/nprobe -- python tools/norm_small_probe.py
Each (shape, variant) is launched 30 times; variants: with / without the dgamma / dbeta atomics."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops

dev = torch.device("cuda", 0)
N = 64
for size, c in ((256, 32), (128, 64), (64, 128), (32, 256), (16, 480)):
    z = torch.randn(N, size, size, c, device=dev).bfloat16()
    g0 = torch.randn(N, size, size, c, device=dev).bfloat16()
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.1
    act = ops.Act(z, ops.instnorm_stats(z, gamma, beta), 0.01)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    for with_params in (True, False):
        for _ in range(30):
            g = g0.clone()
            ops.instnorm_bwd_fused(g, act, gamma, dg if with_params else None, db if with_params else None)
    torch.cuda.synchronize()
print("done")
