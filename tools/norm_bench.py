"""InstanceNorm + LeakyReLU kernels on the training shapes: two-pass (stats + apply / reduce + apply) vs the
resident-chunk one-launch kernels, back to back on one stream (timing experiment, not a test).

    python tools/norm_bench.py [batch] > gpurun_out/norm_bench.txt
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import torch
from cu_hip import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
shapes = [(256, 32), (128, 64), (64, 128), (32, 256), (16, 480), (8, 480), (4, 480), (2, 480)]
REP = 10


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3


print("# size C | fwd: two-pass, resident, grouped, auto us | bwd: two-pass, resident, grouped, auto us | tensor MB")
tot = [0.0] * 8
for size, c in shapes:
    z = torch.randn(N, size, size, c, device=dev).bfloat16()
    g = torch.randn(N, size, size, c, device=dev).bfloat16()
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.1
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)

    def fwd2():
        a = ops.Act(z, ops.instnorm_stats(z, gamma, beta), 0.01)
        ops.instnorm_apply(a)
        return a
    act = fwd2()
    ws = ops._resident_ws(N, c, dev)
    t = [timed(fwd2)] + [timed(lambda m=m: ops.instnorm_fwd_fused(z, gamma, beta, 0.01, ws=ws, mode=m)) for m in (1, 2, 0)]
    failed = ops.resident_wait_failed(ws, N, c)
    t += [timed(lambda: ops.instnorm_lrelu_bwd(g, act, gamma, dg, db, None))]
    for m in (1, 2, 0):
        t.append(timed(lambda m=m: ops.instnorm_bwd_fused(g, act, gamma, dg, db, ws, mode=m)))
        failed = failed or (m == 1 and ops.resident_wait_failed(ws, N, c))
    mb = z.numel() * 2 / 1e6
    print(f"{size:4d} {c:4d} | " + " ".join(f"{v:7.1f}" for v in t[:4]) + " | " + " ".join(f"{v:7.1f}" for v in t[4:]) +
          f" | {mb:7.1f}" + ("  WAIT FAILED" if failed else ""))
    mult = 4 if size > 2 else 2
    for i in range(8):
        tot[i] += mult * t[i]
print("# per step (4 layers per level), ms: fwd " + " ".join(f"{v/1e3:.2f}" for v in tot[:4]) + " | bwd " +
      " ".join(f"{v/1e3:.2f}" for v in tot[4:]))
