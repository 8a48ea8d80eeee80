"""two half-size weight-gradient workgroups per CU on the thin 256^2 layers (igemm_wgrad_dma_kernel<.., IMGB = 39 KiB>,
four waves, 170 VGPRs) against the one-workgroup-per-CU forms.  Tuning build; CU_WGRAD_TWO = 0 / 1 per launch; interleaved rounds
in one process; the partial-tile slabs of both forms are summed and compared.

    CONTOUR_HIP_LIB=$PWD/contouring-uncertainty_amd/libcontour_hip_tuning.so python tools/wgrad_two_bench.py > profiles/r04_wgrad_two_per_cu.txt
"""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from cu_hip.engine import TAPS3_W

DEV = "cuda"
CASES = [(64, 32, 0, 32, 256, "256^2 32->32"), (64, 32, 32, 32, 256, "256^2 32+32->32"), (16, 32, 0, 32, 256, "256^2 32->32, batch 16")]


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rounds, iters = 7, 10
    print("# us per launch, median (min) over %d interleaved rounds of %d launches; algorithmic GB/s = (S + dz bytes) / median" % (rounds, iters))
    for splits in (0, 192):
        print(f"# splits argument {splits} (0 = the library's choice: 256 CUs; 192 = cu_hip.engine's cap for the second stream)")
        for n, c0, c1, co, size, label in CASES:
            dt = torch.bfloat16
            srcs = [ops.Act(torch.randn(n, size, size, c0, device=DEV).to(dt), None, 1.0)]
            if c1:
                srcs.append(ops.Act(torch.randn(n, size, size, c1, device=DEV).to(dt), None, 1.0))
            dz = torch.randn(n, size, size, co, device=DEV).to(dt)
            ws = torch.empty(24 << 20, device=DEV)
            outs, t = [], [[], []]

            def run():
                return ops.conv_wgrad(srcs, dz, ws, grid=(size, size), in_stride=1, z_stride=1, taps=TAPS3_W, n_cols=co, parts=True,
                                      splits=splits)
            for two in (0, 1):
                os.environ["CU_WGRAD_TWO"] = str(two)
                for _ in range(2):
                    slabs = run()
                g = torch.zeros(co, c0 + c1, 3, 3, device=DEV)
                ops.grad_unprep_parts(ws, slabs, co, g, "conv", accumulate=True)
                torch.cuda.synchronize()
                outs.append((g.clone(), slabs[0]))
            for _ in range(rounds):
                for two in (0, 1):
                    os.environ["CU_WGRAD_TWO"] = str(two)
                    t[two].append(timed(run, iters))
            diff = float((outs[0][0] - outs[1][0]).abs().max() / outs[0][0].abs().max())
            m = [statistics.median(x) for x in t]
            nbytes = n * size * size * (c0 + c1 + co) * 2
            print(f"{label:26s} one/CU {m[0]:7.1f} ({min(t[0]):6.1f}) [{outs[0][1]} slabs]   two/CU {m[1]:7.1f} ({min(t[1]):6.1f}) [{outs[1][1]} slabs]"
                  f"   ratio {m[1] / m[0]:5.3f}   {nbytes / m[0] * 1e-3:5.0f} -> {nbytes / m[1] * 1e-3:5.0f} GB/s   max rel diff {diff:.1e}", flush=True)


if __name__ == "__main__":
    main()
