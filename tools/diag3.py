import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd")); sys.path.insert(0, str(ROOT / "tests"))
import torch, torch.nn.functional as F
from cu_hip import ops
DEV = "cuda"
def nhwc(x): return x.permute(0, 2, 3, 1).contiguous()
def nchw(x): return x.permute(0, 3, 1, 2).float()
g = torch.Generator(device=DEV).manual_seed(5)
for (n, c, size) in [(4, 128, 16), (4, 64, 32), (2, 256, 8), (4, 32, 64), (6, 128, 16)]:
    x = torch.randn(n, c, size, size, device=DEV, generator=g) * 2 + 1
    gamma = torch.rand(c, device=DEV, generator=g) + 0.5
    beta = torch.randn(c, device=DEV, generator=g) * 0.2
    z = nhwc(x)
    zf = nchw(z).requires_grad_(True)
    ref = F.leaky_relu(F.instance_norm(zf, weight=gamma, bias=beta, eps=1e-5), 0.01)
    go = torch.randn(n, c, size, size, device=DEV, generator=g)
    ref.backward(go)
    for rep in range(3):
        stats = ops.instnorm_stats(z, gamma, beta, 1e-5)
        gt = nhwc(go)
        dg, db, dbi = (torch.zeros(c, device=DEV) for _ in range(3))
        ops.instnorm_lrelu_bwd(gt, ops.Act(z, stats, 0.01), gamma, dg, db, dbi)
        err = nchw(gt) - zf.grad
        print((n, c, size), rep, "per-image rel err", [f"{float(err[i].norm() / zf.grad[i].norm()):.1e}" for i in range(n)])
