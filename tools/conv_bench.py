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
if len(sys.argv) > 2 and sys.argv[2] == "c64x128":
    cases = [(64, 64, 0, 128, 128, 1)]
if which in ("convT", "s2dgrad"):
    from cu_hip.engine import S2_PARITY_TAPS, _parity_taps
    for (n, ci, co, size) in [(64, 64, 32, 128), (64, 128, 64, 64), (64, 256, 128, 32), (64, 480, 256, 16)]:
        dt = torch.bfloat16
        if which == "convT":        # ConvTranspose2d k2 s2 forward, ci -> co, size -> 2*size, one pass
            x = torch.randn(n, size, size, ci, device=DEV).to(dt)
            w = torch.randn(ci, co, 2, 2, device=DEV) / math.sqrt(ci)
            wf, wd = ops.weight_prep(w, "convT", dt)
            out = torch.empty(n, 2 * size, 2 * size, co, device=DEV, dtype=dt)
            ms = bench(lambda: ops.conv_gemm([ops.Act(x, None, 1.0)], wf.view(1, 4 * co, ci), None, grid=(size, size), in_stride=1,
                                             taps=[(0, 0, 0)], dsts=[out], dst_cols=[co], out_stride=2, n_cols=4 * co, parity_cols=co))
            print(f"convT fwd N={n} C={ci}->{co} {size}x{size}: {ms*1e3:8.1f} us", flush=True)
        else:                       # input gradient of a stride-2 3x3 conv co(=x channels) <- ci(=z channels)
            cx, cz = co, ci
            dz = torch.randn(n, size, size, cz, device=DEV).to(dt)
            w = torch.randn(cz, cx, 3, 3, device=DEV) / math.sqrt(9 * cx)
            _, wd = ops.weight_prep(w, "conv", dt)
            dx = torch.zeros(n, 2 * size, 2 * size, cx, device=DEV, dtype=dt)
            gz = ops.Act(dz, None, 1.0)
            for acc in (0, 1):
                def four():
                    for py in range(2):
                        for px in range(2):
                            taps = [(dy, dx_, kh * 3 + kw) for dy, kh in _parity_taps(py) for dx_, kw in _parity_taps(px)]
                            ops.conv_gemm([gz], wd, None, grid=(size, size), in_stride=1, taps=taps, dsts=[dx], dst_cols=[cx],
                                          out_stride=2, out_off=(py, px), accum=[acc])
                one = lambda: ops.conv_gemm([gz], wd, None, grid=(size, size), in_stride=1,
                                            taps=[(u, v, 0) for u in range(2) for v in range(2)], dsts=[dx], dst_cols=[cx],
                                            out_stride=2, accum=[acc], n_cols=4 * cx, parity_cols=cx, parity_taps=S2_PARITY_TAPS)
                print(f"s2 dgrad N={n} Cz={cz}->Cx={cx} {size}x{size} accum={acc}: four launches {bench(four)*1e3:8.1f} us, one pass {bench(one)*1e3:8.1f} us", flush=True)
    sys.exit(0)
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
