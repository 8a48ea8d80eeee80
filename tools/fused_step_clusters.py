import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd")); sys.path.insert(0, str(ROOT / "tests"))
import torch
import test_head_fused_gpu as T
from contour_uncertainty.data.synthetic import synthetic_batch
torch.manual_seed(0)
task = T._task(stages=6, skew=True)
img, contour = synthetic_batch(4, 64, 21, seed=3)
batch = {"img": img.to(T.DEV), "contour": contour.to(T.DEV)}
def rel(ga, gb):
    out = []
    for n in gb:
        den = float(gb[n].norm())
        if den > 1e-12:
            out.append((float((ga[n] - gb[n]).norm()) / den, n, den))
    return sorted(out, reverse=True)
runs = []
for fused in (True, False, False, True, True, False):
    runs.append((fused, T._grads(task, batch, fused)[1]))
for a in range(len(runs)):
    for b in range(a + 1, len(runs)):
        r = rel(runs[a][1], runs[b][1])
        print(f"run {a} (fused={runs[a][0]}) vs run {b} (fused={runs[b][0]}): worst {r[0][0]:.4f} {r[0][1]} |g|={r[0][2]:.3e}; 2nd {r[1][0]:.4f} {r[1][1]}; median {r[len(r)//2][0]:.5f}")
