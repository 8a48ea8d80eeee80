"""Per-launch table of one training step (timing experiments, not a test): family, shape note, us, TFLOP/s, GB/s.

    python tools/layer_profile.py [batch] [size] [dtype] > gpurun_out/layers.txt
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops
from bench import build_task
from contour_uncertainty.data.synthetic import synthetic_batch

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda", 0)
task, n_stages = build_task(size, dtype, "dsnt-skew")
task = task.to(dev)
opt = task.configure_optimizers()["optimizer"]
img, contour = synthetic_batch(batch, size, 21, seed=1234)
b = {"img": img.to(dev), "contour": contour.to(dev)}


def step(i):
    opt.zero_grad(set_to_none=True)
    out = task.training_step(b, i)
    out["loss"].backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
REP = 3
ops.PROFILE.clear()
ops.PROFILE_ON[0] = True
for i in range(REP):
    step(i)
torch.cuda.synchronize()
ops.PROFILE_ON[0] = False
n = len(ops.PROFILE) // REP
rows = []
for k in range(n):
    fam, flops, _, _, note, nbytes, *_ = ops.PROFILE[k]
    ms = sum(ops.PROFILE[k + r * n][2].elapsed_time(ops.PROFILE[k + r * n][3]) for r in range(REP)) / REP
    rows.append((k, fam, note, ms, flops, nbytes))
tot = sum(r[3] for r in rows)
print(f"# {n} launches/step, {tot:.2f} ms profiled")
print("# idx family note us TFLOP/s GB/s(algorithmic)")
for k, fam, note, ms, flops, nbytes in rows:
    want = sys.argv[4].split(",") if len(sys.argv) > 4 else ("igemm_conv", "igemm_wgrad")
    if fam in want:
        print(f"{k:4d} {fam:12s} {note:44s} {ms*1e3:8.1f} {flops/ms/1e9:7.1f} {nbytes/ms/1e6:7.0f}")
fam_ms = {}
for r in rows:
    fam_ms[r[1]] = fam_ms.get(r[1], 0.0) + r[3]
print("# per family ms:", {k: round(v, 3) for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])})
