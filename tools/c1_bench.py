"""First-layer kernels at the benchmark size (timing experiment): cu_conv_c1_fwd_norm (activation only) and cu_conv_c1_bwd.

    python tools/c1_bench.py [batch] [size]
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
img = torch.rand(n, 1, size, size, device=dev, generator=g)
w = torch.randn(32, 1, 3, 3, device=dev, generator=g)
b = torch.zeros(32, device=dev)
gamma = torch.ones(32, device=dev); beta = torch.zeros(32, device=dev)
w9, _ = ops.weight_prep(w, "conv", torch.float32, want_dgrad=False)
act = ops.conv_c1_fwd_norm(img, w9, b, gamma, beta, 0.01, 1e-5, torch.bfloat16, keep_z=False)
ga = torch.randn(n, size, size, 32, device=dev, generator=g).to(torch.bfloat16)
sums = torch.zeros(n, 32, 2, device=dev); dw = torch.zeros(9, 32, device=dev)
dg = torch.zeros(32, device=dev); db = torch.zeros(32, device=dev)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"c1 fwd (moments + stats + activation): {t(lambda: ops.conv_c1_fwd_norm(img, w9, b, gamma, beta, 0.01, 1e-5, torch.bfloat16, keep_z=False)):.1f} us")
print(f"c1 bwd (two passes):                   {t(lambda: (sums.zero_(), ops.conv_c1_bwd(img, w9, b, act.stats, gamma, 0.01, ga, sums, dw, dg, db))):.1f} us")
