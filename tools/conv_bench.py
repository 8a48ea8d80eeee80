"""Micro-benchmark of single conv_gemm / conv_wgrad launches (timing experiments, not a test)."""
import sys, math, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3, TAPS3_W
DEV = "cuda"
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
cases = [(64, 32, 0, 32, 256, 1), (64, 32, 32, 32, 256, 1), (64, 64, 0, 64, 128, 1), (64, 128, 0, 128, 64, 1), (64, 256, 0, 256, 32, 1),
         (64, 480, 0, 480, 16, 1), (64, 480, 480, 480, 16, 1), (64, 32, 0, 64, 256, 2)]
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
for (n, c0, c1, co, size, stride) in cases:
    dt = torch.bfloat16
    x0 = torch.randn(n, size, size, c0, device=DEV).to(dt)
    st0 = torch.rand(4, n, c0, device=DEV) + 0.5
    plain = os.environ.get('PLAIN', '1') == '1'
    srcs = [ops.Act(x0, None, 1.0) if plain else ops.Act(x0, st0, 0.01)]
    if c1:
        srcs.append(ops.Act(torch.randn(n, size, size, c1, device=DEV).to(dt), None, 1.0))
    ci = c0 + c1
    w = torch.randn(co, ci, 3, 3, device=DEV) / math.sqrt(9 * ci)
    wf, wd = ops.weight_prep(w, "conv", dt)
    os_ = size // stride
    z = torch.empty(n, os_, os_, co, device=DEV, dtype=dt)
    flops = 2.0 * n * os_ * os_ * 9 * ci * co
    if which == "conv":
        ms = bench(lambda: ops.conv_gemm(srcs, wf, None, grid=(os_, os_), in_stride=stride, taps=TAPS3, dsts=[z], dst_cols=[co]))
    else:
        dz = torch.randn(n, os_, os_, co, device=DEV).to(dt)
        dwk = torch.zeros(9, co, ci, device=DEV)
        ms = bench(lambda: ops.conv_wgrad(srcs, dz, dwk, grid=(os_, os_), in_stride=stride, z_stride=1, taps=TAPS3_W, n_cols=co))
    print(f"{which} N={n} C={c0}+{c1}->{co} {size}x{size} s{stride}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s", flush=True)
