"""Diagnostics: per-layer activation-gradient error (da, dz) of the HIP f32 path vs oracle autograd."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd")); sys.path.insert(0, str(ROOT / "tests"))
import torch
from oracle import unet as OU, head as H
from oracle.step import OracleTask, synthetic_batch
from test_model_gpu import make_task

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
img, contour = synthetic_batch(n, 64, 21, seed=99)
ot = OracleTask(spec, task="dsnt-al", seed=3)
taps = {}
logits = OU.unet_forward(ot.sd, img, spec, taps=taps)
for v in taps.values():
    v.retain_grad()
logs = H.dsnt_al_loss(logits, contour)
logs["loss"].backward()
task = make_task("dsnt-al", 6, 64, "f32")
task.model.load_state_dict({k: v.detach() for k, v in ot.sd.items()}, strict=True)
task = task.cuda()
task.model.engine.debug = {}
out = task._shared_step({"img": img.cuda(), "contour": contour.cuda()}, 0)
out["loss"].backward()
dbg = task.model.engine.debug
for k in sorted(dbg, key=lambda s: list(dbg).index(s)):
    prefix, kind = k.split(":")
    ref = taps[f"{prefix}:{'a' if kind == 'da' else 'z'}"].grad
    got = dbg[k].permute(0, 3, 1, 2).cpu()
    err = (got - ref)
    rel = float(err.norm() / ref.norm())
    # where is the error? per-image and border vs interior
    per_img = [float(err[i].norm() / ref[i].norm()) for i in range(err.shape[0])]
    print(f"{k:45s} rel {rel:9.2e}  per-image " + " ".join(f"{v:8.1e}" for v in per_img))

print("---- forward z / stats check per layer (per image): z rel err | mean abs err | rstd rel err | scale rel | shift abs")
ectx_convs = task.model.engine._last_ctx.convs
params = dict(task.model.named_parameters())
for prefix, rec in ectx_convs.items():
    z = rec.out.z.float().permute(0, 3, 1, 2).cpu()
    ref = taps[f"{prefix}:z"].detach()
    st = rec.out.stats.cpu()
    mean = z.mean((2, 3)); var = z.var((2, 3), unbiased=False); rstd = 1 / torch.sqrt(var + 1e-5)
    gam = params[f"{prefix}.norm.weight"].detach().cpu(); bet = params[f"{prefix}.norm.bias"].detach().cpu()
    row = []
    for i in range(z.shape[0]):
        row.append(f"[{float((z[i]-ref[i]).norm()/ref[i].norm()):.0e} {float((st[0,i]-mean[i]).abs().max()):.0e} "
                   f"{float(((st[1,i]-rstd[i])/rstd[i]).abs().max()):.0e} {float(((st[2,i]-gam*rstd[i])/(gam*rstd[i])).abs().max()):.0e} "
                   f"{float((st[3,i]-(bet-mean[i]*gam*rstd[i])).abs().max()):.0e}]")
    print(f"{prefix:34s} " + " ".join(row))
