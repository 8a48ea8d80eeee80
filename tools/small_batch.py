"""Per-GPU batch 8 / 16 / 32 / 64: the eager step (two streams) against the step replayed as one hipGraph.

    python tools/small_batch.py > profiles/r02_small_batch_graph.txt
"""
import json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
print("# python bench.py --batch B --graph on|off --steps 30 --warmup 10 (dsnt-skew 256x256 bf16, 1 GPU)")
print(f"{'batch':>5s} {'eager ms':>10s} {'graph ms':>10s} {'eager img/s':>12s} {'graph img/s':>12s}")
for b in (8, 16, 32, 64):
    row = {}
    for g in ("off", "on"):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--batch", str(b), "--graph", g, "--steps", "30", "--warmup",
                              "10", "--no-cpu-baseline", "--no-roofline", "--no-parity"], capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(f"# batch {b} graph {g}: bench.py failed (rc {out.returncode}): {out.stderr[-400:]!r}", flush=True)
            row[g] = (float("nan"), float("nan"))
            continue
        line = lines[-1]
        d = json.loads(line)
        row[g] = (d["ms_per_step"], d["value"])
    print(f"{b:5d} {row['off'][0]:10.3f} {row['on'][0]:10.3f} {row['off'][1]:12.1f} {row['on'][1]:12.1f}", flush=True)
