"""Timing probe (not a test): the split-K finish that carries the norm (cu_conv_epilogue modes 3 / 4) against the plain
launch + the separate norm launch, on one small-map layer.

    python tools/small_norm_probe.py [size] > gpurun_out/small_norm_probe.txt
"""
import math
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3, TAPS3_D

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
n, c, dt = 64, 480, torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
x = ops.Act(torch.randn(n, size, size, c, device=dev, generator=g).to(dt), None, 1.0)
w = torch.randn(c, c, 3, 3, device=dev, generator=g) / math.sqrt(9 * c)
b = torch.randn(c, device=dev, generator=g)
gamma = torch.rand(c, device=dev, generator=g) + 0.5
beta = torch.randn(c, device=dev, generator=g)
wf, wd = ops.weight_prep(w, "conv", dt)
z = torch.empty(n, size, size, c, device=dev, dtype=dt)
a = torch.empty_like(z)
stats = torch.empty(4, n, c, device=dev)
dgm, dbt = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
tgt = ops.instnorm_fwd_fused(z.normal_(), gamma, beta, 0.01, 1e-5)
d = torch.empty_like(z)


def timeit(name, fn, rep=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:58s} {e0.elapsed_time(e1) / rep * 1e3:8.1f} us")


kw = dict(grid=(size, size), in_stride=1, dst_cols=[c])
print(f"# N{n} {size}x{size} C{c} bf16")
timeit("fwd: conv", lambda: ops.conv_gemm([x], wf, b, taps=TAPS3, dsts=[z], **kw))
timeit("fwd: conv + norm launch", lambda: (ops.conv_gemm([x], wf, b, taps=TAPS3, dsts=[z], **kw),
                                           ops.instnorm_fwd_fused(z, gamma, beta, 0.01, 1e-5)))
timeit("fwd: conv with the norm in its finish", lambda: ops.conv_gemm([x], wf, b, taps=TAPS3, dsts=[z],
                                                                      norm_fwd=(gamma, beta, 1e-5, 0.01, stats, a), **kw))
timeit("bwd: dgrad", lambda: ops.conv_gemm([x], wd, None, taps=TAPS3_D, dsts=[d], **kw))
timeit("bwd: dgrad + norm launch", lambda: (ops.conv_gemm([x], wd, None, taps=TAPS3_D, dsts=[d], **kw),
                                            ops.instnorm_bwd_fused(d, tgt, gamma, dgm, dbt)))
timeit("bwd: dgrad with the norm in its finish", lambda: ops.conv_gemm([x], wd, None, taps=TAPS3_D, dsts=[d],
                                                                       norm_bwd_full=(tgt, gamma, dgm, dbt), **kw))
timeit("bwd: dgrad with the norm in its finish, no dgamma/dbeta", lambda: ops.conv_gemm(
    [x], wd, None, taps=TAPS3_D, dsts=[d], norm_bwd_full=(tgt, gamma, None, None), **kw))
