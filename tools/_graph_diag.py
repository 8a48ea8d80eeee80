"""Diagnostic for tests/test_graph_gpu.py: Adam first moments after 7 steps -- eager vs eager (run-to-run noise of the
f32 atomics) and eager vs hipGraph replay; per-step loss and gradient norm of every run."""
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


def eager_run(tag):
    t = _build(False)
    opt = t.configure_optimizers()["optimizer"]
    for i in range(7):
        opt.zero_grad(set_to_none=True)
        out = t.training_step(batch, i)
        out["loss"].backward()
        g = t.model.last_flat_grad
        print(f"{tag} step {i}: loss {float(out['loss']):.6f} |g| {float(g.norm()):.6e} g[:3] {g[:3].tolist()}", flush=True)
        opt.step()
    return t, opt


def graph_run():
    t = _build(True)
    opt = t.configure_optimizers()["optimizer"]
    st = CapturedStep(t, opt, batch, warmup=3)
    for _ in range(4):
        st.replay()
    st.finish()
    return t, opt


a, oa = eager_run("A")
b, ob = eager_run("B")
d, od = eager_run("D")
c, oc = graph_run()
names = [n for n, _ in a.model.named_parameters()]
pa, pb, pc, pd = (list(t.model.parameters()) for t in (a, b, c, d))
print(f"{'parameter':50s} {'|m|':>10s} {'A-B':>10s} {'A-D':>10s} {'B-D':>10s} {'A-graph':>10s}")
for n, x, y, z, w in list(zip(names, pa, pb, pc, pd))[:12]:
    if x not in oa.state:
        continue
    ma, mb, mc, md = oa.state[x]["exp_avg"], ob.state[y]["exp_avg"], oc.state[z]["exp_avg"], od.state[w]["exp_avg"]
    nm = float(ma.norm()) + 1e-30
    print(f"{n:50s} {nm:10.3e} {float((ma - mb).norm()) / nm:10.4f} {float((ma - md).norm()) / nm:10.4f} "
          f"{float((mb - md).norm()) / nm:10.4f} {float((ma - mc).norm()) / nm:10.4f}")
