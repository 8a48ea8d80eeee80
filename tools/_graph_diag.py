import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd")); sys.path.insert(0, str(ROOT / "tests"))
import torch
from test_graph_gpu import _build
from contour_uncertainty.data.synthetic import synthetic_batch
from cu_hip.graph import CapturedStep
img, contour = synthetic_batch(4, 64, 21, seed=3)
batch = {"img": img.cuda(), "contour": contour.cuda()}
def eager(cap):
    t = _build(cap); opt = t.configure_optimizers()["optimizer"]; ls = []
    for i in range(7):
        opt.zero_grad(set_to_none=True); out = t.training_step(batch, i); out["loss"].backward(); opt.step(); ls.append(round(float(out["loss"]), 4))
    return ls
print("eager A ", eager(False)); print("eager B ", eager(False)); print("eager capturable-adam", eager(True))
t = _build(True); o = t.configure_optimizers()["optimizer"]
st = CapturedStep(t, o, batch, warmup=3); g = []
for _ in range(4):
    st.replay(); g.append(round(float(st.logs["loss"]), 4))
print("graph   ", g)
