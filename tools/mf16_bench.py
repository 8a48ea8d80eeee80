"""v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 in the wide-layer ring kernels
(igemm_conv_dma_ring_kernel<NB, NXR, MF16>), same tile, same LDS images, same DMA schedule.  Needs the tuning build
(make -C contouring-uncertainty_amd/csrc TUNING=1; CONTOUR_HIP_LIB=.../libcontour_hip_tuning.so): CU_CONV_MF16 = 0 / 1 picks the form
per launch.  Interleaved rounds in ONE process on random data (cdna_hip_programming.md rule 24 / 25); the outputs of the two
forms are compared (same bf16 inputs, f32 accumulation in a different order).

    CONTOUR_HIP_LIB=$PWD/contouring-uncertainty_amd/libcontour_hip_tuning.so python tools/mf16_bench.py > profiles/r04_mf16_ring.txt
"""
import math
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3, TAPS3_D

DEV = "cuda"
# (N, C0, C1, CO, size, label): forward convs and input gradients of the 16^2 ... 128^2 levels at batch 64
CASES = [(64, 128, 0, 128, 64, "64^2 128->128"), (64, 128, 128, 128, 64, "64^2 128+128->128"), (64, 128, 0, 256, 64, "64^2 128->256 (dgrad, 2 dst)"),
         (64, 256, 0, 256, 32, "32^2 256->256"), (64, 256, 256, 256, 32, "32^2 256+256->256"), (64, 256, 0, 512, 32, "32^2 256->512 (dgrad, 2 dst)"),
         (64, 480, 0, 480, 16, "16^2 480->480"), (64, 480, 480, 480, 16, "16^2 480+480->480"), (64, 64, 64, 64, 128, "128^2 64+64->64")]


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rounds, iters = 7, 20
    print("# us per launch, median (min) over %d interleaved rounds of %d launches; TFLOP/s from the median" % (rounds, iters))
    print("# layer                          32x32x16        16x16x32        ratio   max |diff| / max |out|")
    tot = [0.0, 0.0]
    for n, c0, c1, co, size, label in CASES:
        dt = torch.bfloat16
        srcs = [ops.Act(torch.randn(n, size, size, c0, device=DEV).to(dt), None, 1.0)]
        if c1:
            srcs.append(ops.Act(torch.randn(n, size, size, c1, device=DEV).to(dt), None, 1.0))
        w = torch.randn(co, c0 + c1, 3, 3, device=DEV) / math.sqrt(9 * (c0 + c1))
        wf, _ = ops.weight_prep(w, "conv", dt)
        two = "2 dst" in label
        dsts = [torch.empty(n, size, size, co // 2 if two else co, device=DEV, dtype=dt) for _ in range(2 if two else 1)]
        cols = [d.shape[3] for d in dsts]
        bias = None if two else torch.randn(co, device=DEV)
        taps = TAPS3_D if two else TAPS3
        fn = lambda: ops.conv_gemm(srcs, wf, bias, grid=(size, size), in_stride=1, taps=taps, dsts=dsts, dst_cols=cols)
        outs, t = [], [[], []]
        for mf in (0, 1):
            os.environ["CU_CONV_MF16"] = str(mf)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            outs.append([d.float().clone() for d in dsts])
        for _ in range(rounds):
            for mf in (0, 1):
                os.environ["CU_CONV_MF16"] = str(mf)
                t[mf].append(timed(fn, iters))
        diff = max(float((a - b).abs().max()) for a, b in zip(*outs)) / max(float(a.abs().max()) for a in outs[0])
        m = [statistics.median(x) for x in t]
        fl = 2.0 * n * size * size * 9 * (c0 + c1) * co
        tot[0] += m[0]; tot[1] += m[1]
        print(f"{label:30s} {m[0]:7.1f} ({min(t[0]):6.1f}) {m[1]:7.1f} ({min(t[1]):6.1f})   {m[1] / m[0]:5.3f}   {diff:.1e}"
              f"   {fl / m[0] * 1e-6:6.0f} -> {fl / m[1] * 1e-6:6.0f} TFLOP/s", flush=True)
    print(f"# sum {tot[0]:.1f} -> {tot[1]:.1f} us  ({tot[1] / tot[0]:.3f})")


if __name__ == "__main__":
    main()
