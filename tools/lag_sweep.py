"""Round 4 experiment: weight-gradient launches trailing the input-gradient chain by CONTOUR_WGRAD_LAG layers (cu_hip/engine.py),
optionally with another CU cap for the second stream (CONTOUR_WGRAD_WGS).  Whole-step time of bench.py (batch 64, bf16, eager),
every configuration twice, interleaved, on one box.

    python tools/lag_sweep.py > profiles/r04_wgrad_lag_sweep.txt
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
configs = [(0, 192), (2, 192), (4, 192), (6, 192), (8, 192), (12, 192), (20, 192), (100, 192), (6, 256), (6, 128), (12, 256), (0, 256)]
if len(sys.argv) > 1:
    configs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
print("# python bench.py --steps 40 --warmup 10 (dsnt-skew 256x256 bf16, batch 64, 1 GPU): ms per step, two rounds")
res = {c: [] for c in configs}
for rnd in range(2):
    for lag, wgs in configs:
        env = dict(os.environ, CONTOUR_WGRAD_LAG=str(lag), CONTOUR_WGRAD_WGS=str(wgs))
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "40", "--warmup", "10", "--no-cpu-baseline",
                              "--no-roofline", "--no-parity"], capture_output=True, text=True, env=env)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res[(lag, wgs)].append(json.loads(lines[-1])["ms_per_step"] if lines else float("nan"))
        print(f"#   round {rnd} lag {lag:3d} wgs {wgs:3d}: {res[(lag, wgs)][-1]:.3f}", flush=True)
print(f"{'lag':>4s} {'wgs':>4s} {'ms (1)':>8s} {'ms (2)':>8s}")
for (lag, wgs), v in res.items():
    print(f"{lag:4d} {wgs:4d} " + " ".join(f"{x:8.3f}" for x in v))
