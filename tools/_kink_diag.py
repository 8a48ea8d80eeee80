"""diagnostic (not a test): are there pixels exactly on the LeakyReLU kink in test_thin_streaming_conv case 1?"""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd")); sys.path.insert(0, str(ROOT / "tests"))
import torch, torch.nn.functional as F
import test_kernels_gpu as T
ops = T._ops()
DEV = T.DEV
n, c0, c1, co, size = (16, 32, 32, 32, 256)
dtype = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(31)
for c in (c0, c1):
    T.make_act(torch.randn(n, c, size, size, device=DEV, generator=g), dtype, False, 1.0, g)
ci = c0 + c1
w = torch.randn(co, ci, 3, 3, device=DEV, generator=g) / math.sqrt(9 * ci)
b = torch.randn(co, device=DEV, generator=g) * 0.1
gm_ = 1 + 0.1 * torch.randn(co, device=DEV, generator=g)
bt_ = 0.1 * torch.randn(co, device=DEV, generator=g)
wq = T.rq(w, dtype)
dz = T.rq(torch.randn(n, co, size, size, device=DEV, generator=g), dtype)
gx = F.conv_transpose2d(dz, wq, None, padding=1)
zt = T.rq(torch.randn(n, ci, size, size, device=DEV, generator=g), dtype)
gm = 1 + 0.1 * torch.randn(ci, device=DEV, generator=g)
bt = 0.1 * torch.randn(ci, device=DEV, generator=g)
tgt = ops.Act(T.nhwc(zt, dtype), None, 0.01)
for rep in range(4):
    tgt.stats = ops.instnorm_stats(tgt.z, gm, bt)
    mean, rstd, scale, shift = (tgt.stats[i][:, :, None, None] for i in range(4))
    y = zt * scale + shift
    amb = y.abs() <= 4e-7 * (zt.abs() * scale.abs() + shift.abs())
    cnt = amb.sum((2, 3))
    nz = cnt.nonzero()
    print("rep", rep, "maps with on-kink pixels:", [(int(a), int(b_), int(cnt[a, b_]), round(float((amb[a, b_] * gx[a, b_]).sum()), 3), float(y[a, b_][amb[a, b_]].sum())) for a, b_ in nz.tolist()][:6], flush=True)
