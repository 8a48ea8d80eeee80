"""Fused head kernels at the benchmark size (timing experiment): cu_head_fused_fwd / _bwd, 64 x 256 x 256 x 32 bf16, K = 21.

    python tools/head_bench.py        (tuning build: CU_HF_ROWS, CU_HF_WGS)
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops

n, size, k = 64, 256, 21
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
z = (torch.randn(n, size, size, 32, device=dev, generator=g) * 1.5).to(torch.bfloat16)
gamma = torch.ones(32, device=dev); beta = torch.zeros(32, device=dev)
act = ops.instnorm_fwd_fused(z, gamma, beta, 0.01, materialize=False)
w = torch.randn(k, 32, 1, 1, device=dev, generator=g) * 0.5
w_cls, w_ch = ops.weight_prep(w, "conv", torch.bfloat16, 32)
mu, sg, aux = ops.head_fused_fwd(act, w_cls, k, True)
gmu = torch.randn(n, k, 2, device=dev, generator=g) * 0.1
gsg = torch.randn(n, k, 3, device=dev, generator=g) * 0.01
sums = torch.zeros(n, 32, 2, device=dev)
parts = torch.empty(1025 * 1024, device=dev)


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


print(f"head fwd {t(lambda: ops.head_fused_fwd(act, w_cls, k, True)):.1f} us   head bwd {t(lambda: ops.head_fused_bwd(act, w_cls, w_ch, k, aux, gmu, gsg, True, sums, parts)):.1f} us")
