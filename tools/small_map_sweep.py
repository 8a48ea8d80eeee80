"""the <= 8x8 half of the network, knob by knob (tuning build).  One ConvLayer forward = the split-K convolution +
its finish pass, which carries InstanceNorm + LeakyReLU (cu_conv_gemm_ex epilogue mode 3); timed as the engine issues it, with
true device durations summed per call from HIP events around a burst of 20 calls.

Knobs (csrc/igemm_conv.hip): CU_CONV_NO_NARROW (keep the widest column tile: more weight reuse, fewer workgroups),
CU_CONV_KSPLIT_TARGET (workgroups the channel split aims at, default 512), CU_CONV_KSPLIT_MAX (largest split, default 8),
CU_CONV_KSPLIT_WGS (largest un-split grid that still gets a split, default 128), CU_CONV_DMA_MINWG (fewest workgroups for the 8-wave
LDS-DMA kernel, default 128 / 64).

    CONTOUR_HIP_LIB=$PWD/contouring-uncertainty_amd/libcontour_hip_tuning.so python tools/small_map_sweep.py > profiles/r04_small_map_sweep.txt
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
from cu_hip.engine import TAPS3

DEV = "cuda"
KNOBS = ["CU_CONV_NO_NARROW", "CU_CONV_KSPLIT_TARGET", "CU_CONV_KSPLIT_MAX", "CU_CONV_KSPLIT_WGS", "CU_CONV_DMA_MINWG", "CU_CONV_NBMAX"]
VARIANTS = [("default", {}), ("wide tiles", {"CU_CONV_NO_NARROW": "1"}), ("target 256", {"CU_CONV_KSPLIT_TARGET": "256"}),
            ("target 1024, max 15", {"CU_CONV_KSPLIT_TARGET": "1024", "CU_CONV_KSPLIT_MAX": "15"}),
            ("wide + target 1024, max 15", {"CU_CONV_NO_NARROW": "1", "CU_CONV_KSPLIT_TARGET": "1024", "CU_CONV_KSPLIT_MAX": "15"}),
            ("64-column tiles", {"CU_CONV_NBMAX": "2"}), ("64-column tiles, target 1024", {"CU_CONV_NBMAX": "2", "CU_CONV_KSPLIT_TARGET": "1024", "CU_CONV_KSPLIT_MAX": "15"}),
            ("target 128", {"CU_CONV_KSPLIT_TARGET": "128"}), ("target 256, max 4", {"CU_CONV_KSPLIT_TARGET": "256", "CU_CONV_KSPLIT_MAX": "4"}),
            ("wide + target 256", {"CU_CONV_NO_NARROW": "1", "CU_CONV_KSPLIT_TARGET": "256"}),
            ("LDS-DMA kernel from 16 workgroups", {"CU_CONV_DMA_MINWG": "16"})]


def timed(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n, dt = 64, torch.bfloat16
    print("# us per ConvLayer forward (conv + norm-carrying finish), batch 64, median of 5 interleaved rounds; every variant's output is "
          "compared with the default's")
    for size in (8, 4, 2):
        for ci in (480, 960):
            srcs = [ops.Act(torch.randn(n, size, size, 480, device=DEV).to(dt), None, 1.0) for _ in range(ci // 480)]
            w = torch.randn(480, ci, 3, 3, device=DEV) / math.sqrt(9 * ci)
            wf, _ = ops.weight_prep(w, "conv", dt)
            bias = torch.randn(480, device=DEV) * 0.1
            gamma, beta = torch.rand(480, device=DEV) + 0.5, torch.randn(480, device=DEV) * 0.1
            z = torch.empty(n, size, size, 480, device=DEV, dtype=dt)
            stats = torch.empty(4, n, 480, device=DEV)
            act = torch.empty_like(z)

            def layer():
                got = ops.conv_gemm(srcs, wf, bias, grid=(size, size), in_stride=1, taps=TAPS3, dsts=[z], dst_cols=[480],
                                    norm_fwd=(gamma, beta, 1e-5, 0.01, stats, act))
                if not got:
                    ops.instnorm_fwd_fused(z, gamma, beta, 0.01, 1e-5)
                return got
            t = {name: [] for name, _ in VARIANTS}
            ref, took = None, {}
            for name, env in VARIANTS:
                for k in KNOBS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                took[name] = layer()
                torch.cuda.synchronize()
                if ref is None:
                    ref = z.float().clone()
                else:
                    err = float((z.float() - ref).abs().max() / ref.abs().max())
                    if err >= 2e-2:
                        took[name] = f"WRONG RESULT (max rel diff {err:.2f})"
            for _ in range(5):
                for name, env in VARIANTS:
                    for k in KNOBS:
                        os.environ.pop(k, None)
                    os.environ.update(env)
                    t[name].append(timed(layer))
            base = statistics.median(t["default"])
            print(f"{size}x{size} C{ci}->480:")
            for name, _ in VARIANTS:
                m = statistics.median(t[name])
                tag = "" if took[name] is True else ("   [separate norm launch]" if took[name] is False else "   " + took[name])
                print(f"    {name:36s} {m:7.1f} us  ({m / base:5.2f} x){tag}", flush=True)


if __name__ == "__main__":
    main()
