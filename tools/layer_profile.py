"""Per-launch table of one training step (timing experiments, not a test): family, shape note, us, TFLOP/s, GB/s, and the
launch's FLOOR = max(algorithmic FLOPs / 1.4 PFLOP/s, algorithmic bytes / 5.5 TB/s) -- what a tuned bf16 MFMA loop on random data
and a streaming kernel reach on this chip (MI355X_MICROARCH.md: 1.3-1.5 PFLOP/s; 6.3 TB/s copy, ~5.5 with a halo) -- so that the
"where the step stands" sums of DESIGN.md can be checked launch by launch.

    CONTOUR_SIDE_WGRAD=0 python tools/layer_profile.py [batch] [size] [dtype] [families | all] > profiles/rNN_layers_one_stream.txt
(one stream: an event pair then times one launch; with the weight-gradient stream beside it, it times both streams' kernels)
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
print("# idx family note us TFLOP/s GB/s(algorithmic) floor_us")
PF, TB = 1.4e15, 5.5e12
floor_tot, level = 0.0, {}
for k, fam, note, ms, flops, nbytes in rows:
    want = sys.argv[4].split(",") if len(sys.argv) > 4 else ("all",)
    floor = max(flops / PF, nbytes / TB) * 1e6
    floor_tot += floor if floor > 0 else ms * 1e3          # launches without a work model count at their measured time
    import re
    m = re.search(r"N\d+ (\d+)x\d+", note)
    lv = (m.group(1) + "^2") if m else "other"
    a = level.setdefault(lv, [0.0, 0.0, 0])
    a[0] += ms * 1e3; a[1] += floor if floor > 0 else ms * 1e3; a[2] += 1
    if "all" in want or fam in want:
        print(f"{k:4d} {fam:12s} {note:44s} {ms*1e3:8.1f} {flops/ms/1e9:7.1f} {nbytes/ms/1e6:7.0f} {floor:8.1f}")
print(f"# floor sum {floor_tot / 1e3:.2f} ms of {tot:.2f} ms measured (launches without a FLOP / byte model at their measured time)")
print("# by map size: launches, measured us, floor us")
for lv, (a0, a1, c) in sorted(level.items(), key=lambda kv: -kv[1][0]):
    print(f"#   {lv:8s} {c:4d} {a0:9.1f} {a1:9.1f}")
fam_ms = {}
for r in rows:
    fam_ms[r[1]] = fam_ms.get(r[1], 0.0) + r[3]
print("# per family ms:", {k: round(v, 3) for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])})
