"""GPU parity tests, kernel by kernel, through the C ABI (cu_hip.ops) against PyTorch fp32 references of the same op.

f32 mode uses the exact-f32 MFMA path: tolerance 1e-4 relative (BASELINE.json north_star).
bf16 mode is fed bf16-representable inputs and compared with the fp32 reference of the SAME rounded inputs, so the only
differences are accumulation order and the final rounding of the output to bf16 (2^-8 relative).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from cu_hip import ops
    return ops


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 1.2e-2


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def nchw(x):
    return x.permute(0, 3, 1, 2).float()


def rq(x, dtype):
    """round to dtype and back (so that the fp32 reference sees the same values)."""
    return x.to(dtype).float()


def make_act(x_nchw, dtype, with_norm, slope, gen):
    """Returns (Act, reference activated tensor NCHW fp32)."""
    ops = _ops()
    z = nhwc(x_nchw, dtype)
    zf = nchw(z)
    if with_norm:
        n, c = x_nchw.shape[:2]
        stats = torch.zeros(4, n, c, device=DEV)
        stats[2] = torch.rand(n, c, device=DEV, generator=gen) + 0.5
        stats[3] = torch.randn(n, c, device=DEV, generator=gen) * 0.3
        a = zf * stats[2][:, :, None, None] + stats[3][:, :, None, None]
    else:
        stats = None
        a = zf
    a = torch.where(a > 0, a, a * slope)
    if dtype == torch.bfloat16:
        a = rq(a, dtype)           # the kernel rounds the activated operand to bf16 before the MFMA
    return ops.Act(z, stats, slope), a


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [
    # n, cin0, cin1, cout, size, stride
    (2, 32, 0, 32, 64, 1),
    (3, 32, 32, 64, 32, 1),      # concat
    (2, 64, 0, 128, 32, 2),      # stride 2
    (5, 480, 0, 480, 4, 1),      # multi-image tile, 480 channels (BN = 96)
    (4, 256, 0, 480, 4, 2),      # stride 2 on a tiny map (image count per tile is reduced)
    (2, 480, 480, 480, 8, 1),
    (64, 480, 0, 128, 2, 1),     # ConfidenceNet-like 2x2
    (64, 480, 480, 480, 4, 1),   # bottom of the U at batch 64: split-K over the channel chunks (cu_conv_gemm_ws)
    (64, 480, 0, 480, 2, 2),
])
def test_conv_fwd(dtype, case):
    ops = _ops()
    from cu_hip.engine import TAPS3
    n, c0, c1, co, size, stride = case
    g = torch.Generator(device=DEV).manual_seed(1)
    x0 = torch.randn(n, c0, size, size, device=DEV, generator=g)
    a0, r0 = make_act(x0, dtype, True, 0.01, g)
    srcs, refs = [a0], [r0]
    if c1:
        x1 = torch.randn(n, c1, size, size, device=DEV, generator=g)
        a1, r1 = make_act(x1, dtype, False, 1.0, g)
        srcs.append(a1)
        refs.append(r1)
    w = torch.randn(co, c0 + c1, 3, 3, device=DEV, generator=g) / math.sqrt(9 * (c0 + c1))
    b = torch.randn(co, device=DEV, generator=g) * 0.1
    wf, _ = ops.weight_prep(w, "conv", dtype)
    wq = rq(w, dtype)
    os_ = size // stride
    z = torch.empty(n, os_, os_, co, device=DEV, dtype=dtype)
    ops.conv_gemm(srcs, wf, b, grid=(os_, os_), in_stride=stride, taps=TAPS3, dsts=[z], dst_cols=[co])
    ref = F.conv2d(torch.cat(refs, 1), wq, b, stride=stride, padding=1)
    assert rel_err(nchw(z), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 64, 32, 64, 1), (2, 64, 128, 32, 2), (3, 960, 480, 4, 2), (2, 128, 256, 16, 1),
                                  (64, 960, 480, 4, 1), (64, 512, 480, 2, 1)])      # tiny maps at batch 64: split-K
def test_conv_dgrad_wgrad(dtype, case):
    """input gradient (incl. concat split + accumulate) and weight gradient vs autograd of F.conv2d."""
    ops = _ops()
    from cu_hip.engine import TAPS3_D, TAPS3_W, _parity_taps
    n, ci, co, size, stride = case
    g = torch.Generator(device=DEV).manual_seed(2)
    c0 = ci // 2
    xa = torch.randn(n, c0, size, size, device=DEV, generator=g)
    xb = torch.randn(n, ci - c0, size, size, device=DEV, generator=g)
    a0, r0 = make_act(xa, dtype, True, 0.01, g)
    a1, r1 = make_act(xb, dtype, False, 1.0, g)
    w = torch.randn(co, ci, 3, 3, device=DEV, generator=g) / math.sqrt(9 * ci)
    wq = rq(w, dtype).requires_grad_(True)
    xin = torch.cat([r0, r1], 1).requires_grad_(True)
    os_ = size // stride
    dz = rq(torch.randn(n, co, os_, os_, device=DEV, generator=g), dtype)
    out = F.conv2d(xin, wq, None, stride=stride, padding=1)
    out.backward(dz)
    # ---- dgrad into two destinations, the second one accumulating onto existing data
    _, wd = ops.weight_prep(w, "conv", dtype)
    d0 = torch.empty(n, size, size, c0, device=DEV, dtype=dtype)
    base = rq(torch.randn(n, ci - c0, size, size, device=DEV, generator=g), dtype)
    d1 = nhwc(base, dtype)
    gz = ops.Act(nhwc(dz, dtype), None, 1.0)
    if stride == 1:
        ops.conv_gemm([gz], wd, None, grid=(size, size), in_stride=1, taps=TAPS3_D, dsts=[d0, d1],
                      dst_cols=[c0, ci - c0], accum=[0, 1])
    else:
        for py in range(2):
            for px in range(2):
                taps = [(dy, dx, kh * 3 + kw) for dy, kh in _parity_taps(py) for dx, kw in _parity_taps(px)]
                ops.conv_gemm([gz], wd, None, grid=(size // 2, size // 2), in_stride=1, taps=taps, dsts=[d0, d1],
                              dst_cols=[c0, ci - c0], out_stride=2, out_off=(py, px), accum=[0, 1])
    assert rel_err(nchw(d0), xin.grad[:, :c0]) < tol(dtype)
    assert rel_err(nchw(d1), xin.grad[:, c0:] + base) < tol(dtype)
    # ---- wgrad
    dwk = torch.zeros(9, co, ci, device=DEV)
    ops.conv_wgrad([a0, a1], gz.z, dwk, grid=(os_, os_), in_stride=stride, z_stride=1, taps=TAPS3_W, n_cols=co)
    gw = torch.zeros_like(w)
    ops.grad_unprep(dwk, gw, "conv", accumulate=True)
    assert rel_err(gw, wq.grad) < (2e-4 if dtype == torch.float32 else 1e-2)
    # ---- the same through partial tiles (cu_conv_wgrad_parts + cu_grad_unprep_parts): poisoned scratch, no atomics,
    #      bit-identical run to run; once with the plain (materialised) operand the engine feeds (LDS-DMA kernel)
    for srcs in ([a0, a1], [ops.Act(nhwc(r0, dtype), None, 1.0), ops.Act(nhwc(r1, dtype), None, 1.0)]):
        outs = []
        for _ in range(2):
            ws = torch.full((24 << 20,), float("nan"), device=DEV)
            slabs = ops.conv_wgrad(srcs, gz.z, ws, grid=(os_, os_), in_stride=stride, z_stride=1, taps=TAPS3_W, n_cols=co,
                                    parts=True)
            assert slabs[0] >= 1
            gp = torch.zeros_like(w)
            ops.grad_unprep_parts(ws, slabs, co, gp, "conv", accumulate=True)
            outs.append(gp)
        assert rel_err(outs[0], wq.grad) < (2e-4 if dtype == torch.float32 else 1e-2)
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 64, 32, 16), (3, 480, 480, 2), (2, 480, 256, 8), (4, 128, 64, 32), (3, 256, 128, 16)])
def test_conv_transpose(dtype, case):
    ops = _ops()
    n, ci, co, size = case
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(n, ci, size, size, device=DEV, generator=g)
    a, r = make_act(x, dtype, True, 0.01, g)
    w = torch.randn(ci, co, 2, 2, device=DEV, generator=g) / math.sqrt(ci)
    wq = rq(w, dtype).requires_grad_(True)
    rin = r.clone().requires_grad_(True)
    ref = F.conv_transpose2d(rin, wq, None, stride=2)
    wf, wd = ops.weight_prep(w, "convT", dtype)
    u = torch.empty(n, 2 * size, 2 * size, co, device=DEV, dtype=dtype)
    for dy in range(2):
        for dx in range(2):
            ops.conv_gemm([a], wf, None, grid=(size, size), in_stride=1, taps=[(0, 0, dy * 2 + dx)], dsts=[u],
                          dst_cols=[co], out_stride=2, out_off=(dy, dx))
    assert rel_err(nchw(u), ref) < tol(dtype)
    du = rq(torch.randn(n, co, 2 * size, 2 * size, device=DEV, generator=g), dtype)
    ref.backward(du)
    dun = nhwc(du, dtype)
    din = torch.empty(n, size, size, ci, device=DEV, dtype=dtype)
    ops.conv_gemm([ops.Act(dun, None, 1.0)], wd, None, grid=(size, size), in_stride=2,
                  taps=[(dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)], dsts=[din], dst_cols=[ci])
    assert rel_err(nchw(din), rin.grad) < tol(dtype)
    dwk = torch.zeros(4, co, ci, device=DEV)
    a_plain = ops.Act(nhwc(r, dtype), None, 1.0)      # materialised activation (what the engine feeds)
    ops.conv_wgrad([a_plain], dun, dwk, grid=(size, size), in_stride=1, z_stride=2,
                   taps=[(0, 0, dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)], n_cols=co)
    gw = torch.zeros_like(w)
    ops.grad_unprep(dwk, gw, "convT", accumulate=True)
    assert rel_err(gw, wq.grad) < (2e-4 if dtype == torch.float32 else 1e-2)
    ws = torch.full((24 << 20,), float("nan"), device=DEV)          # partial-tile form
    slabs = ops.conv_wgrad([a_plain], dun, ws, grid=(size, size), in_stride=1, z_stride=2,
                            taps=[(0, 0, dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)], n_cols=co, parts=True)
    gp = torch.zeros_like(w)
    ops.grad_unprep_parts(ws, slabs, co, gp, "convT", accumulate=True)
    assert rel_err(gp, wq.grad) < (2e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(3, 64, 64), (40, 40, 24), (64, 40, 56)])
def test_first_conv(dtype, geom):
    """Cin = 1 layer, forward and weight gradient (row chunks of 1, 2 and 3 rows, the last chunk partial)."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(4)
    n, hh, ww = geom
    co = 32
    img = torch.rand(n, 1, hh, ww, device=DEV, generator=g)
    w = torch.randn(co, 1, 3, 3, device=DEV, generator=g)
    b = torch.randn(co, device=DEV, generator=g)
    w9, _ = ops.weight_prep(w, "conv", torch.float32, want_dgrad=False)
    z = torch.empty(n, hh, ww, co, device=DEV, dtype=dtype)
    ops.conv_c1_fwd(img, w9, b, z)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(img, wr, b, padding=1)
    assert rel_err(nchw(z), ref) < tol(dtype)
    dz = rq(torch.randn(n, co, hh, ww, device=DEV, generator=g), dtype)
    ref.backward(dz)
    dw9 = torch.zeros(9, co, device=DEV)
    ops.conv_c1_wgrad(img, nhwc(dz, dtype), dw9)
    gw = torch.zeros_like(w)
    ops.grad_unprep(dw9.view(9, co, 1), gw, "conv", accumulate=True)
    assert rel_err(gw, wr.grad) < 2e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(3, 64, 64, 32), (5, 256, 256, 32), (2, 32, 128, 16), (2, 19, 45, 8), (1, 3, 5, 32)])
def test_first_conv_with_norm(dtype, geom):
    """cu_conv_c1_fwd_norm (statistics from 54 moments of the image, z and a written by one pass) against the three-launch
    form it replaces and against F.instance_norm of the unrounded conv: z bit-identical, statistics to f32 rounding, a to
    one rounding step of the storage type; bit-identical run to run."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(14)
    n, hh, ww, co = geom
    img = torch.rand(n, 1, hh, ww, device=DEV, generator=g) + 0.5          # a mean well above the spread: cancellation
    w = torch.randn(co, 1, 3, 3, device=DEV, generator=g)
    w[0] -= w[0].mean()                                                     # an edge filter: output mean ~ 0
    b = torch.randn(co, device=DEV, generator=g) * 3
    gamma = torch.rand(co, device=DEV, generator=g) + 0.5
    beta = torch.randn(co, device=DEV, generator=g)
    w9, _ = ops.weight_prep(w, "conv", torch.float32, want_dgrad=False)
    z = torch.empty(n, hh, ww, co, device=DEV, dtype=dtype)
    ops.conv_c1_fwd(img, w9, b, z)
    new = ops.conv_c1_fwd_norm(img, w9, b, gamma, beta, 0.01, 1e-5, dtype)
    again = ops.conv_c1_fwd_norm(img, w9, b, gamma, beta, 0.01, 1e-5, dtype)
    assert torch.equal(new.z, z)
    assert torch.equal(new.stats, again.stats) and torch.equal(new.a, again.a)
    zd = F.conv2d(img.double(), w.double(), b.double(), padding=1)         # the unrounded conv output
    mean = zd.mean((2, 3))
    rstd = (zd.var((2, 3), unbiased=False) + 1e-5).rsqrt()
    assert rel_err(new.stats[0], mean.float()) < 2e-6
    assert rel_err(new.stats[1], rstd.float()) < 2e-5
    old = ops.instnorm_fwd_fused(z, gamma, beta, 0.01, 1e-5)               # statistics of the ROUNDED z
    assert rel_err(new.stats, old.stats) < (1e-4 if dtype == torch.float32 else 2e-3)
    step = 2.0 ** -7 if dtype == torch.bfloat16 else 0.0
    d = (new.a.float() - old.a.float()).abs()
    # (a 15-pixel image does not average the rounding of z away: the two sets of statistics differ by ~1e-3 there)
    assert (d <= 2 * step * old.a.float().abs() + (5e-3 if dtype == torch.bfloat16 else 1e-4) * old.a.float().abs().max()).all()
    ref = F.leaky_relu(F.instance_norm(nchw(z).float(), weight=gamma, bias=beta, eps=1e-5), 0.01)
    assert rel_err(nchw(new.a), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(3, 64, 64, 32), (4, 256, 256, 32), (2, 32, 128, 32), (2, 19, 45, 32), (1, 3, 5, 64)])
def test_first_layer_backward_without_z(dtype, geom):
    """cu_conv_c1_bwd (z recomputed from the image, dz never stored; two passes over dL/da) against autograd through
    conv -> instance_norm -> leaky_relu on the same image, and against the chain it replaces (norm backward in place, then
    cu_conv_c1_wgrad); the activation of the a-only forward is bit-identical to the z-keeping one."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(24)
    n, hh, ww, co = geom
    img = torch.rand(n, 1, hh, ww, device=DEV, generator=g) + 0.5
    w = torch.randn(co, 1, 3, 3, device=DEV, generator=g)
    b = torch.randn(co, device=DEV, generator=g)
    gamma = torch.rand(co, device=DEV, generator=g) + 0.5
    beta = torch.randn(co, device=DEV, generator=g) * 0.5
    w9, _ = ops.weight_prep(w, "conv", torch.float32, want_dgrad=False)
    full = ops.conv_c1_fwd_norm(img, w9, b, gamma, beta, 0.01, 1e-5, dtype)
    lean = ops.conv_c1_fwd_norm(img, w9, b, gamma, beta, 0.01, 1e-5, dtype, keep_z=False)
    assert torch.equal(full.a, lean.a) and torch.equal(full.stats, lean.stats) and lean.z is lean.a
    ga = rq(torch.randn(n, co, hh, ww, device=DEV, generator=g), dtype)
    g_nhwc = nhwc(ga, dtype)
    # ---- the chain it replaces
    dz = g_nhwc.clone()
    dgam0, dbet0 = torch.zeros(co, device=DEV), torch.zeros(co, device=DEV)
    ops.instnorm_bwd_fused(dz, full, gamma, dgam0, dbet0)
    dw0 = torch.zeros(9, co, device=DEV)
    ops.conv_c1_wgrad(img, dz, dw0)
    # ---- fused
    sums = torch.zeros(n, co, 2, device=DEV)
    dw1 = torch.zeros(9, co, device=DEV)
    dgam1, dbet1 = torch.zeros(co, device=DEV), torch.zeros(co, device=DEV)
    ops.conv_c1_bwd(img, w9, b, lean.stats, gamma, 0.01, g_nhwc, sums, dw1, dgam1, dbet1)
    torch.cuda.synchronize()
    loose = dtype == torch.bfloat16           # the chain rounds dz to bf16 before the weight gradient, the fused form does not
    assert rel_err(dw1, dw0) < (2e-2 if loose else 2e-4)
    assert rel_err(dgam1, dgam0) < 2e-4 and rel_err(dbet1, dbet0) < 2e-4
    # ---- autograd on the stored (rounded) z's statistics is not what the kernels use (moments of the unrounded conv):
    #      f32 only, where the two coincide
    if dtype == torch.float32 and hh * ww >= 1024:
        wr, gr_, br = w.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        out = F.leaky_relu(F.instance_norm(F.conv2d(img, wr, b, padding=1), weight=gr_, bias=br, eps=1e-5), 0.01)
        out.backward(ga)
        gw = torch.zeros_like(w)
        ops.grad_unprep(dw1.view(9, co, 1), gw, "conv", accumulate=True)
        assert rel_err(gw, wr.grad) < 1e-3 and rel_err(dgam1, gr_.grad) < 1e-3 and rel_err(dbet1, br.grad) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 32, 64), (3, 480, 4), (2, 128, 16), (64, 480, 2)])
def test_instnorm_fwd_bwd(dtype, shape):
    """statistics + fused backward vs F.instance_norm -> leaky_relu autograd."""
    ops = _ops()
    n, c, size = shape
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, c, size, size, device=DEV, generator=g) * 2 + 3.0      # |mean| > std on purpose
    gamma = torch.rand(c, device=DEV, generator=g) + 0.5
    beta = torch.randn(c, device=DEV, generator=g) * 0.2
    z = nhwc(x, dtype)
    zf = nchw(z).requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    ref = F.leaky_relu(F.instance_norm(zf, weight=gm, bias=bt, eps=1e-5), 0.01)
    stats = ops.instnorm_stats(z, gamma, beta, 1e-5)
    mean = zf.detach().mean((2, 3))
    var = zf.detach().var((2, 3), unbiased=False)
    assert torch.allclose(stats[0], mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(stats[1], 1 / torch.sqrt(var + 1e-5), rtol=2e-4)
    act = ops.Act(z, stats, 0.01)
    out = ops.act_to_nchw_f32(act)
    assert rel_err(out, ref.detach()) < 2e-4
    go = rq(torch.randn(n, c, size, size, device=DEV, generator=g), dtype)
    ref.backward(go)
    gt = nhwc(go, dtype)
    dgamma = torch.zeros(c, device=DEV)
    dbeta = torch.zeros(c, device=DEV)
    dbias = torch.zeros(c, device=DEV)
    ops.instnorm_lrelu_bwd(gt, act, gamma, dgamma, dbeta, dbias)
    t = 2e-4 if dtype == torch.float32 else 1.5e-2
    assert rel_err(nchw(gt), zf.grad) < t
    assert rel_err(dgamma, gm.grad) < 2e-4 and rel_err(dbeta, bt.grad) < 2e-4
    # conv bias in front of an InstanceNorm has an exactly-zero gradient up to rounding
    assert float(dbias.abs().max()) < 1e-2 * float(go.abs().sum() / c) + 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 32, 128), (2, 32, 64), (3, 480, 4), (2, 128, 16), (64, 480, 2), (5, 64, 48),
                                   (2, 256, 32), (3, 480, 16), (2, 480, 8)])
@pytest.mark.parametrize("mode", [1, 2, 0])
def test_instnorm_fused_entry_points(dtype, shape, mode):
    """cu_instnorm_fwd_fused / cu_instnorm_bwd_fused in their resident-chunk form (mode 1: one launch, every tensor read
    once), their grouped two-pass form (mode 2) and as the engine calls them (mode 0) vs F.instance_norm -> leaky_relu
    autograd: images spanning 1 .. 256 chunks, a ragged last chunk (48 x 48), 480 channels (threads-per-pixel not a power
    of two), tiny maps."""
    ops = _ops()
    n, c, size = shape
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, c, size, size, device=DEV, generator=g) * 2 + 3.0      # |mean| > std on purpose
    gamma = torch.rand(c, device=DEV, generator=g) + 0.5
    beta = torch.randn(c, device=DEV, generator=g) * 0.2
    z = nhwc(x, dtype)
    zf = nchw(z).requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    ref = F.leaky_relu(F.instance_norm(zf, weight=gm, bias=bt, eps=1e-5), 0.01)
    act = ops.instnorm_fwd_fused(z, gamma, beta, 0.01, 1e-5, mode=mode)
    assert mode != 1 or not ops.resident_wait_failed(act.ws, n, c)
    mean = zf.detach().mean((2, 3))
    var = zf.detach().var((2, 3), unbiased=False)
    assert torch.allclose(act.stats[0], mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(act.stats[1], 1 / torch.sqrt(var + 1e-5), rtol=2e-4)
    assert torch.allclose(act.stats[2], gamma[None] * act.stats[1], rtol=1e-6)
    assert rel_err(nchw(act.a), ref.detach()) < (2e-4 if dtype == torch.float32 else 1e-2)
    # identical to the two-pass kernels
    stats2 = ops.instnorm_stats(z, gamma, beta, 1e-5)
    assert torch.allclose(stats2, act.stats, rtol=1e-5, atol=1e-6)
    go = rq(torch.randn(n, c, size, size, device=DEV, generator=g), dtype)
    ref.backward(go)
    gt = nhwc(go, dtype)
    dgamma = torch.zeros(c, device=DEV)
    dbeta = torch.zeros(c, device=DEV)
    ws = ops.instnorm_bwd_fused(gt, act, gamma, dgamma, dbeta, mode=mode)
    assert mode != 1 or not ops.resident_wait_failed(ws, n, c)
    t = 2e-4 if dtype == torch.float32 else 1.5e-2
    # LeakyReLU' is discontinuous at 0: an element whose normalised value is within rounding noise of 0 may be decided
    # differently by two correct implementations (one such element among the 1.5 M of the 128 x 128 case); bound their
    # number, hold everything else to the tolerance
    d = (nchw(gt) - zf.grad).abs() / zf.grad.abs().max()
    assert int((d > t).sum()) <= 2 and float(d[d <= t].max()) < t
    assert rel_err(dgamma, gm.grad) < 5e-3 and rel_err(dbeta, bt.grad) < 5e-3      # a kink element moves the sums


@pytest.mark.parametrize("size", [16, 64, 256])
def test_dsnt_head_vs_golden(golden_dir, size):
    ops = _ops()
    g = np.load(golden_dir / "dsnt_head.npz")
    tag = f"s{size}"
    logits = torch.from_numpy(g[f"{tag}_logits"]).to(DEV)
    n, k = logits.shape[:2]
    mu, sigma, aux = ops.dsnt_head_fwd(logits, True)
    half = size / 2
    assert torch.allclose(mu.cpu(), torch.from_numpy(g[f"{tag}_pixel"]), rtol=0, atol=2e-4)
    assert torch.allclose(sigma[..., :2].cpu(), torch.from_numpy(g[f"{tag}_var"]) * half ** 2, rtol=1e-4, atol=1e-6)
    assert torch.allclose(sigma[..., 2].cpu(), torch.from_numpy(g[f"{tag}_covar"]) * half ** 2, rtol=1e-4, atol=1e-5)
    # backward: golden upstream grads are w.r.t. normalised coords/var/covar -> convert to pixel-unit grads
    gmu = torch.from_numpy(g[f"{tag}_g_coords"]).to(DEV) / (0.5 * size)
    gs = torch.cat([torch.from_numpy(g[f"{tag}_g_var"]), torch.from_numpy(g[f"{tag}_g_covar"])[..., None]], -1)
    gs = (gs / half ** 2).to(DEV)
    dl = ops.dsnt_head_bwd(logits, aux, gmu.contiguous(), gs.contiguous(), True).cpu()
    if size <= 64:
        ref = torch.from_numpy(g[f"{tag}_dlogits"])
        assert rel_err(dl, ref) < 1e-4
    else:
        ref = torch.from_numpy(g[f"{tag}_dlogits_row"])
        assert float((dl[0, 0, 100] - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    # the same gradient in the layout / element type the UNet's backward stages (cu_dsnt_head_bwd_nhwc): f32 is the same
    # arithmetic up to fused-multiply-add contraction, bf16 its rounding; channels K.. are zero
    d32 = ops.dsnt_head_bwd_nhwc(logits, aux, gmu.contiguous(), gs.contiguous(), True, torch.float32).cpu()
    assert d32.shape == (n, size, size, 32) and float(d32[..., k:].abs().max()) == 0.0
    scale = float(dl.abs().max())
    assert float((d32[..., :k].permute(0, 3, 1, 2) - dl).abs().max()) <= 2e-6 * scale
    d16 = ops.dsnt_head_bwd_nhwc(logits, aux, gmu.contiguous(), gs.contiguous(), True, torch.bfloat16).cpu()
    err = (d16[..., :k].permute(0, 3, 1, 2).float() - dl).abs()
    assert bool((err <= dl.abs() * 2.0 ** -8 + 2e-6 * scale).all()) and float(d16[..., k:].abs().max()) == 0.0


def test_nll_vs_golden(golden_dir):
    ops = _ops()
    g = np.load(golden_dir / "nll_heads.npz")
    mu = torch.from_numpy(g["mu"]).squeeze(-1).to(DEV).contiguous()
    y = torch.from_numpy(g["y"]).squeeze(-1).to(DEV).contiguous()
    cov = torch.from_numpy(g["cov"]).to(DEV)
    alpha = torch.from_numpy(g["alpha"]).squeeze(-1).to(DEV).contiguous()
    sig3 = torch.stack([cov[:, 0, 0], cov[:, 1, 1], cov[:, 0, 1]], -1).contiguous()
    m = mu.shape[0]
    # Gaussian (dsnt_al.py:64-71)
    logs, gmu, gsig, _ = ops.nll_fwd_bwd(mu, sig3, y, None)
    ref = float(g["gauss_loss"])
    assert abs(float(logs[0]) - ref) < 1e-4 * abs(ref)
    assert abs(float(logs[2]) - float(g["gauss_t1_mean"])) < 1e-4 * abs(float(g["gauss_t1_mean"]))
    assert rel_err(gmu.cpu(), torch.from_numpy(g["gauss_dmu"]).squeeze(-1)) < 1e-4
    dcov = torch.from_numpy(g["gauss_dcov"])
    ref3 = torch.stack([dcov[:, 0, 0], dcov[:, 1, 1], dcov[:, 0, 1] + dcov[:, 1, 0]], -1)
    assert rel_err(gsig.cpu(), ref3) < 2e-4
    # skew (bivariateskewnormal.py:51-61)
    logs, gmu, gsig, gal = ops.nll_fwd_bwd(mu, sig3, y, alpha)
    nll_ref = torch.from_numpy(g["skew_nll"])
    assert abs(float(logs[0]) - float(nll_ref.mean())) < 1e-4 * abs(float(nll_ref.mean()))
    assert abs(float(logs[4]) - float(g["skew_t3"].mean())) < 1e-4 * abs(float(g["skew_t3"].mean())) + 1e-6
    # 0.5*(1+erf(z/sqrt2)) cancels in fp32 once Phi(z) is small: the reference's own value carries ~6e-8/Phi relative
    # rounding noise there, so rows with Phi < 1e-2 are compared at 2e-2 and the well-conditioned rows at 2e-4.
    cdf = torch.from_numpy(np.exp(g["skew_t3"]) - 1e-7)
    well = cdf > 1e-2
    for got, key in ((gmu, "skew_dmu"), (gal, "skew_dalpha")):
        ref_ = torch.from_numpy(g[key]).squeeze(-1)
        assert rel_err(got.cpu()[well], ref_[well]) < 2e-4, key
        assert rel_err(got.cpu()[~well], ref_[~well]) < 2e-2, key
    dcov = torch.from_numpy(g["skew_dcov"])
    ref3 = torch.stack([dcov[:, 0, 0], dcov[:, 1, 1], dcov[:, 0, 1] + dcov[:, 1, 0]], -1)
    ok = torch.isfinite(ref3).all(-1)      # linalg.eig backward is undefined at repeated eigenvalues
    ok[1] = False                          # near-isotropic row: eig backward is ill-conditioned in the reference
    ok &= well
    assert rel_err(gsig.cpu()[ok], ref3[ok]) < 5e-3


def test_adam_matches_torch():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(6)
    n = 100003
    p = torch.randn(n, device=DEV, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=1e-3)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(n, device=DEV, generator=g)
        ref.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-3, step)
    assert torch.allclose(p, ref.detach(), rtol=1e-5, atol=1e-6)


def test_linear(golden_dir):
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(5, 512, device=DEV, generator=g)
    w = torch.randn(42, 512, device=DEV, generator=g) * 0.05
    b = torch.randn(42, device=DEV, generator=g)
    out = ops.linear_fwd(x, w, b)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.linear(xr, wr, br)
    assert rel_err(out, ref) < 1e-5
    go = torch.randn(5, 42, device=DEV, generator=g)
    ref.backward(go)
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    gx = ops.linear_bwd(x, w, go, gw, gb)
    assert rel_err(gx, xr.grad) < 1e-5 and rel_err(gw, wr.grad) < 1e-5 and rel_err(gb, br.grad) < 1e-5


@pytest.mark.parametrize("case", [
    # n, cin0, cin1, cout, size, (d0 = split of the destination columns or None), accum of the second destination
    (64, 256, 0, 256, 32, None, 0),       # encoder 32x32
    (40, 128, 128, 128, 64, None, 0),     # decoder concat at 64x64, ragged image count
    (64, 480, 0, 480, 16, None, 0),       # 480 columns: last 128-column tile is partial; two images per tile
    (64, 480, 0, 960, 16, 480, 1),        # input gradient of a decoder conv: two destinations, the second accumulated
    (32, 64, 0, 128, 128, 64, 0),         # 64 input channels (two chunks), two destinations of 64 columns
    (32, 128, 0, 192, 64, None, 0),       # 192 columns: the 64-column variant (two taps per staging round)
    (6, 32, 0, 32, 256, None, 0),         # thin layers on the same kernel: 32-column variant (four taps per staging round)
    (5, 32, 32, 32, 256, None, 0),        # ... decoder concat 32 + 32 -> 32
    (6, 32, 0, 64, 256, 32, 1),           # ... its input gradient: two destinations of 32 columns, the second accumulated
    (12, 64, 64, 64, 128, None, 0),       # 128 x 128 level
])
def test_conv_wide_dma_kernel(case):
    """bf16 3x3 stride-1 layers with >= 128 channels take the 8-wave LDS-DMA kernel (igemm_conv_dma_kernel): compare
    with F.conv2d and with the 4-wave register-staged kernel (CU_CONV_NODMA=1), which sums in the same order."""
    import os
    ops = _ops()
    from cu_hip.engine import TAPS3
    n, c0, c1, co, size, d0, acc1 = case
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(3)
    srcs, refs = [], []
    for c in (c0, c1):
        if c:
            x = torch.randn(n, c, size, size, device=DEV, generator=g)
            a, r = make_act(x, dtype, False, 1.0, g)
            srcs.append(a)
            refs.append(r)
    w = torch.randn(co, c0 + c1, 3, 3, device=DEV, generator=g) / math.sqrt(9 * (c0 + c1))
    b = torch.randn(co, device=DEV, generator=g) * 0.1
    wf, _ = ops.weight_prep(w, "conv", dtype)
    ref = F.conv2d(torch.cat(refs, 1), rq(w, dtype), b, padding=1)

    def run():
        if d0 is None:
            z = torch.empty(n, size, size, co, device=DEV, dtype=dtype)
            ops.conv_gemm(srcs, wf, b, grid=(size, size), in_stride=1, taps=TAPS3, dsts=[z], dst_cols=[co])
            return [z]
        za = torch.empty(n, size, size, d0, device=DEV, dtype=dtype)
        zb = torch.ones(n, size, size, co - d0, device=DEV, dtype=dtype)
        ops.conv_gemm(srcs, wf, b, grid=(size, size), in_stride=1, taps=TAPS3, dsts=[za, zb], dst_cols=[d0, co - d0],
                      accum=(0, acc1))
        return [za, zb]

    out = run()
    os.environ["CU_CONV_NODMA"] = "1"
    try:
        old = run()
    finally:
        del os.environ["CU_CONV_NODMA"]
    for a, o in zip(out, old):
        assert torch.equal(a, o)
    got = torch.cat([nchw(t) for t in out], 1)
    if d0 is not None and acc1:
        ref = ref.clone()
        ref[:, d0:] += 1.0
    assert rel_err(got, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose_single_pass_matches_four_parity_launches(dtype):
    """parity-column mode of cu_conv_gemm (W[4][CO][CI] viewed as 4*CO GEMM columns, one pass over the source) vs the
    four per-parity launches and vs F.conv_transpose2d."""
    ops = _ops()
    n, ci, co, size = 3, 64, 32, 16
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, ci, size, size, device=DEV, generator=g)
    a, r = make_act(x, dtype, False, 1.0, g)
    w = torch.randn(ci, co, 2, 2, device=DEV, generator=g) / math.sqrt(ci)
    wf, _ = ops.weight_prep(w, "convT", dtype)
    one = torch.empty(n, 2 * size, 2 * size, co, device=DEV, dtype=dtype)
    ops.conv_gemm([a], wf.view(1, 4 * co, ci), None, grid=(size, size), in_stride=1, taps=[(0, 0, 0)], dsts=[one],
                  dst_cols=[co], out_stride=2, n_cols=4 * co, parity_cols=co)
    four = torch.empty_like(one)
    for dy in range(2):
        for dx in range(2):
            ops.conv_gemm([a], wf, None, grid=(size, size), in_stride=1, taps=[(0, 0, dy * 2 + dx)], dsts=[four],
                          dst_cols=[co], out_stride=2, out_off=(dy, dx))
    # same products, but the single pass may split its channel chunks over workgroups (cu_conv_gemm_ws): the f32 sums
    # then differ in their last bits (and a bf16 result by one rounding step)
    assert rel_err(one.float(), four.float()) < (1e-6 if dtype == torch.float32 else 2e-3)
    ref = F.conv_transpose2d(r, rq(w, dtype), stride=2)
    assert rel_err(nchw(one), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_weight_prep_batch_matches_per_layer(dtype):
    """cu_weight_prep_batch (every layer in one launch) vs cu_weight_prep, conv and transposed-conv layouts, padded COP."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(7)
    layers = [(torch.randn(64, 32, 3, 3, device=DEV, generator=g), "conv", None),
              (torch.randn(480, 256, 3, 3, device=DEV, generator=g), "conv", None),
              (torch.randn(40, 24, 3, 3, device=DEV, generator=g), "conv", None),
              (torch.randn(128, 64, 2, 2, device=DEV, generator=g), "convT", None),
              (torch.randn(24, 40, 2, 2, device=DEV, generator=g), "convT", None),
              (torch.randn(21, 32, 1, 1, device=DEV, generator=g), "conv", 32)]
    entries, outs = [], []
    for w, kind, cop in layers:
        t, co, ci, _, _ = ops._layout(w.shape, kind)
        cp = cop or co
        wf = torch.full((t, cp, ci), 7.0, device=DEV).to(dtype)
        wd = torch.full((t, ci, cp), 7.0, device=DEV).to(dtype)
        entries.append((w, wf, wd, kind, cop))
        outs.append((wf, wd))
    table, blocks = ops.prep_table(entries, torch.device(DEV))
    ops.weight_prep_batch(table, len(entries), blocks, dtype)
    for (w, kind, cop), (wf, wd) in zip(layers, outs):
        rf, rd = ops.weight_prep(w, kind, dtype, cop)
        assert torch.equal(wf, rf) and torch.equal(wd, rd)


@pytest.mark.parametrize("case", [(64, 64, 128, 64), (40, 128, 256, 32), (64, 256, 512, 16), (64, 256, 480, 32)])
def test_conv_wide_dma_kernel_stride2(case):
    """stride-2 3x3 forward on the 8-wave LDS-DMA kernel (256-pixel tiles, 17 x 65 halo) vs the register-staged kernel
    (same summation order) and F.conv2d."""
    import os
    ops = _ops()
    from cu_hip.engine import TAPS3
    n, ci, co, size = case
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(n, ci, size, size, device=DEV, generator=g)
    a, r = make_act(x, dtype, False, 1.0, g)
    w = torch.randn(co, ci, 3, 3, device=DEV, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, device=DEV, generator=g) * 0.1
    wf, _ = ops.weight_prep(w, "conv", dtype)
    os_ = size // 2

    def run():
        z = torch.empty(n, os_, os_, co, device=DEV, dtype=dtype)
        ops.conv_gemm([a], wf, b, grid=(os_, os_), in_stride=2, taps=TAPS3, dsts=[z], dst_cols=[co])
        return z

    out = run()
    os.environ["CU_CONV_NODMA"] = "1"
    try:
        old = run()
    finally:
        del os.environ["CU_CONV_NODMA"]
    assert torch.equal(out, old)
    ref = F.conv2d(r, rq(w, dtype), b, stride=2, padding=1)
    assert rel_err(nchw(out), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 32, 64, 32, 0), (2, 64, 128, 16, 1), (3, 480, 480, 4, 1), (64, 480, 480, 2, 0),
                                  (2, 256, 480, 8, 1)])
def test_stride2_dgrad_single_pass(dtype, case):
    """Input gradient of a stride-2 3x3 conv with all four input parities in one launch (par_taps) vs autograd of
    F.conv2d, with and without accumulation onto existing data."""
    ops = _ops()
    from cu_hip.engine import S2_PARITY_TAPS
    n, ci, co, os_, accum = case
    size = 2 * os_
    g = torch.Generator(device=DEV).manual_seed(11)
    w = torch.randn(co, ci, 3, 3, device=DEV, generator=g) / math.sqrt(9 * ci)
    wq = rq(w, dtype)
    xin = torch.zeros(n, ci, size, size, device=DEV, requires_grad=True)
    dz = rq(torch.randn(n, co, os_, os_, device=DEV, generator=g), dtype)
    F.conv2d(xin, wq, None, stride=2, padding=1).backward(dz)
    _, wd = ops.weight_prep(w, "conv", dtype)
    base = rq(torch.randn(n, ci, size, size, device=DEV, generator=g), dtype)
    d0 = nhwc(base, dtype)
    gz = ops.Act(nhwc(dz, dtype), None, 1.0)
    ops.conv_gemm([gz], wd, None, grid=(os_, os_), in_stride=1, taps=[(u, v, 0) for u in range(2) for v in range(2)],
                  dsts=[d0], dst_cols=[ci], out_stride=2, accum=[accum], n_cols=4 * ci, parity_cols=ci,
                  parity_taps=S2_PARITY_TAPS)
    ref = xin.grad + (base if accum else 0)
    assert rel_err(nchw(d0), ref) < tol(dtype)


@pytest.mark.parametrize("case", [(4, 64, 32, 64), (3, 128, 64, 128), (16, 480, 256, 16), (8, 256, 128, 32),
                                  (64, 480, 256, 16), (16, 256, 128, 32)])      # >= 16384 loop pixels: the 256 x 256 tile form
def test_lean_gather_gemm_shapes(case):
    """The shapes cu_conv_gemm hands to the lean gather-GEMM (pconv.hip; bf16, >= 4096 / 16384 loop pixels): transposed
    conv forward (one pass, parity scatter), its input gradient (4 taps, stride-2 gather), and the stride-2 3x3 input
    gradient in one pass (9 of 16 (tap, parity) pairs) with and without accumulation -- vs autograd of the PyTorch ops.
    480 channels = 7.5 K-chunks (zero-filled tail), 256 / 480 output columns = several / ragged column tiles."""
    ops = _ops()
    from cu_hip.engine import S2_PARITY_TAPS
    n, ci, co, size = case
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(21)
    # ---- transposed conv forward + input gradient
    x = torch.randn(n, ci, size, size, device=DEV, generator=g)
    a, r = make_act(x, dtype, False, 1.0, g)
    w = torch.randn(ci, co, 2, 2, device=DEV, generator=g) / math.sqrt(ci)
    wq = rq(w, dtype)
    rin = r.clone().requires_grad_(True)
    ref = F.conv_transpose2d(rin, wq, None, stride=2)
    wf, wd = ops.weight_prep(w, "convT", dtype)
    u = torch.empty(n, 2 * size, 2 * size, co, device=DEV, dtype=dtype)
    ops.conv_gemm([a], wf.view(1, 4 * co, ci), None, grid=(size, size), in_stride=1, taps=[(0, 0, 0)], dsts=[u],
                  dst_cols=[co], out_stride=2, n_cols=4 * co, parity_cols=co)
    assert rel_err(nchw(u), ref) < tol(dtype)
    du = rq(torch.randn(n, co, 2 * size, 2 * size, device=DEV, generator=g), dtype)
    ref.backward(du)
    base = rq(torch.randn(n, ci, size, size, device=DEV, generator=g), dtype)
    for accum in (0, 1):
        din = nhwc(base, dtype)
        ops.conv_gemm([ops.Act(nhwc(du, dtype), None, 1.0)], wd, None, grid=(size, size), in_stride=2,
                      taps=[(dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)], dsts=[din], dst_cols=[ci],
                      accum=[accum])
        assert rel_err(nchw(din), rin.grad + (base if accum else 0)) < tol(dtype)
    # ---- stride-2 3x3 conv (cx -> cz): input gradient in one pass
    cx, cz, os_ = co, ci, size
    w3 = torch.randn(cz, cx, 3, 3, device=DEV, generator=g) / math.sqrt(9 * cx)
    xin = torch.zeros(n, cx, 2 * os_, 2 * os_, device=DEV, requires_grad=True)
    dz = rq(torch.randn(n, cz, os_, os_, device=DEV, generator=g), dtype)
    F.conv2d(xin, rq(w3, dtype), None, stride=2, padding=1).backward(dz)
    _, wd3 = ops.weight_prep(w3, "conv", dtype)
    base = rq(torch.randn(n, cx, 2 * os_, 2 * os_, device=DEV, generator=g), dtype)
    gz = ops.Act(nhwc(dz, dtype), None, 1.0)
    for accum in (0, 1):
        d0 = nhwc(base, dtype)
        ops.conv_gemm([gz], wd3, None, grid=(os_, os_), in_stride=1, taps=[(u_, v_, 0) for u_ in range(2) for v_ in range(2)],
                      dsts=[d0], dst_cols=[cx], out_stride=2, accum=[accum], n_cols=4 * cx, parity_cols=cx,
                      parity_taps=S2_PARITY_TAPS)
        assert rel_err(nchw(d0), xin.grad + (base if accum else 0)) < tol(dtype)


@pytest.mark.parametrize("case", [(16, 32, 0, 32, 256), (16, 32, 32, 32, 256), (64, 64, 0, 64, 128), (64, 64, 0, 32, 128),
                                  (20, 32, 0, 32, 256)])
def test_thin_streaming_conv(case):
    """The streaming kernel of the thin, large layers (tconv.hip; bf16, >= 2^20 loop pixels): forward (+ bias), concat
    forward, and the input gradient into one and into two destinations -- vs F.conv2d and its autograd.  20 images x 8192
    tiles is not a multiple of 8 workgroup strides: ragged tail of the persistent loop (out-of-range DMA instructions)."""
    ops = _ops()
    from cu_hip.engine import TAPS3, TAPS3_D
    n, c0, c1, co, size = case
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(31)
    srcs, refs = [], []
    for c in (c0, c1):
        if c:
            a, r = make_act(torch.randn(n, c, size, size, device=DEV, generator=g), dtype, False, 1.0, g)
            srcs.append(a)
            refs.append(r)
    ci = c0 + c1
    w = torch.randn(co, ci, 3, 3, device=DEV, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, device=DEV, generator=g) * 0.1
    wf, wd = ops.weight_prep(w, "conv", dtype)
    wq = rq(w, dtype)
    z = torch.empty(n, size, size, co, device=DEV, dtype=dtype)
    sums = torch.zeros(n, co, 2, device=DEV)
    got = ops.conv_gemm(srcs, wf, b, grid=(size, size), in_stride=1, taps=TAPS3, dsts=[z], dst_cols=[co], stat_sums=sums)
    ref = F.conv2d(torch.cat(refs, 1), wq, b, padding=1)
    assert rel_err(nchw(z), ref) < tol(dtype)
    # the InstanceNorm statistics gathered in the epilogue (cu_conv_gemm_stats): sum / sum of squares of (z - bias) per
    # (image, channel), from the f32 accumulators; then the norm's forward from them vs F.instance_norm of the f32 output
    assert got == (n != 20)           # 20 images: 10 tiles per workgroup do not divide an image (256 tiles) -> plain launch
    if got:
        c0_ = ref - b[None, :, None, None]
        s1, s2 = c0_.sum((2, 3)), (c0_ * c0_).sum((2, 3))
        assert rel_err(sums[..., 0], s1) < 2e-3 and rel_err(sums[..., 1], s2) < 2e-3
        gm = 1 + 0.1 * torch.randn(co, device=DEV, generator=g)
        bt = 0.1 * torch.randn(co, device=DEV, generator=g)
        act = ops.instnorm_fwd_given(z, gm, bt, 0.01, sums, b)
        want = F.leaky_relu(F.instance_norm(nchw(z), weight=gm, bias=bt, eps=1e-5), 0.01)
        assert rel_err(nchw(act.a), want) < 2e-2
    # borders: the zero padding comes from the buffer range check
    assert rel_err(nchw(z)[:, :, [0, -1]], ref[:, :, [0, -1]]) < tol(dtype)
    assert rel_err(nchw(z)[:, :, :, [0, -1]], ref[:, :, :, [0, -1]]) < tol(dtype)
    del ref
    # ---- input gradient: dz (n, co) -> dx (n, ci), into one destination and split into two
    dz = rq(torch.randn(n, co, size, size, device=DEV, generator=g), dtype)
    gx = F.conv_transpose2d(dz, wq, None, padding=1)
    gz = ops.Act(nhwc(dz, dtype), None, 1.0)
    if ci in (32, 64) and co in (32, 64):
        # ... with the reduction pass of the TARGET layer's InstanceNorm backward in the epilogue (cu_conv_gemm_ex mode 2):
        # the launch produces g = dL/da of a layer a = LeakyReLU(scale z + shift) and the sums of gl and gl * zhat
        zt = rq(torch.randn(n, ci, size, size, device=DEV, generator=g), dtype)
        gm = 1 + 0.1 * torch.randn(ci, device=DEV, generator=g)
        bt = 0.1 * torch.randn(ci, device=DEV, generator=g)
        tgt = ops.Act(nhwc(zt, dtype), None, 0.01)
        tgt.stats = ops.instnorm_stats(tgt.z, gm, bt)
        sums = torch.zeros(n, ci, 2, device=DEV)
        d0 = torch.empty(n, size, size, ci, device=DEV, dtype=dtype)
        got = ops.conv_gemm([gz], wd, None, grid=(size, size), in_stride=1, taps=TAPS3_D, dsts=[d0], dst_cols=[ci],
                            norm_bwd=(tgt, sums))
        assert rel_err(nchw(d0), gx) < tol(dtype)
        assert got == (n != 20)
        if got:
            mean, rstd, scale, shift = (tgt.stats[i][:, :, None, None] for i in range(4))
            y = zt * scale + shift
            gl = torch.where(y > 0, gx, gx * 0.01)
            zhat = (zt - mean) * rstd
            # z is bf16: when -shift / scale of an (image, channel) falls on a representable value, every pixel holding that
            # value has y = 0 to within the rounding of its two terms, and its LeakyReLU branch hangs on the last bit of the
            # statistics (atomics: run to run) and on fma against mul + add.  Those pixels' contributions are slack.
            amb = (y.abs() <= 4e-7 * (zt.abs() * scale.abs() + shift.abs())).float()
            for j, ref_j, w_j in ((0, gl.sum((2, 3)), gx.abs()), (1, (gl * zhat).sum((2, 3)), (gx * zhat).abs())):
                slack = (amb * w_j).sum((2, 3))
                assert bool(((sums[..., j] - ref_j).abs() <= 5e-3 * ref_j.abs().max() + slack).all())
            # the apply pass from those sums == the two-pass backward
            g1, g2 = d0.clone(), d0.clone()
            dg1, db1, dg2, db2 = (torch.zeros(ci, device=DEV) for _ in range(4))
            ops.instnorm_bwd_given(g1, tgt, gm, dg1, db1, sums)
            ops.instnorm_lrelu_bwd(g2, tgt, gm, dg2, db2, None)
            assert rel_err(g1.float(), g2.float()) < 1e-2 and rel_err(dg1, dg2) < 1e-2 and rel_err(db1, db2) < 1e-2
    if ci == 64 and co == 32:
        da = torch.empty(n, size, size, 32, device=DEV, dtype=dtype)
        db = torch.empty(n, size, size, 32, device=DEV, dtype=dtype)
        ops.conv_gemm([gz], wd, None, grid=(size, size), in_stride=1, taps=TAPS3_D, dsts=[da, db], dst_cols=[32, 32])
        assert rel_err(nchw(da), gx[:, :32]) < tol(dtype) and rel_err(nchw(db), gx[:, 32:]) < tol(dtype)


@pytest.mark.parametrize("case", [(3, 64, 128, 32), (2, 32, 64, 64), (4, 256, 480, 16), (2, 128, 256, 64), (5, 480, 480, 16),
                                  (2, 64, 128, 8)])
def test_stride2_wgrad_parity_planes(case):
    """Weight gradient of a stride-2 3x3 conv from ONE plain bf16 source (what the engine launches for every downsampling
    block): the LDS-DMA kernel stages the four parity planes of the source (dense k-loop).  Ragged image counts, 16- and
    32-pixel rows, 8-pixel rows (the general k-step addressing), the 480-channel tail; atomics and partial-tile forms."""
    ops = _ops()
    from cu_hip.engine import TAPS3_W
    n, ci, co, size = case
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(21)
    x = rq(torch.randn(n, ci, size, size, device=DEV, generator=g), dtype).requires_grad_(True)
    w = (torch.randn(co, ci, 3, 3, device=DEV, generator=g) / math.sqrt(9 * ci)).requires_grad_(True)
    dz = rq(torch.randn(n, co, size // 2, size // 2, device=DEV, generator=g), dtype)
    F.conv2d(x, w, None, stride=2, padding=1).backward(dz)
    src = ops.Act(nhwc(x.detach(), dtype), None, 1.0)
    zt = nhwc(dz, dtype)
    os_ = size // 2
    dwk = torch.zeros(9, co, ci, device=DEV)
    ops.conv_wgrad([src], zt, dwk, grid=(os_, os_), in_stride=2, z_stride=1, taps=TAPS3_W, n_cols=co)
    gw = torch.zeros_like(w)
    ops.grad_unprep(dwk, gw, "conv", accumulate=True)
    assert rel_err(gw, w.grad) < 2e-3
    ws = torch.full((24 << 20,), float("nan"), device=DEV)
    slabs = ops.conv_wgrad([src], zt, ws, grid=(os_, os_), in_stride=2, z_stride=1, taps=TAPS3_W, n_cols=co, parts=True)
    gp = torch.zeros_like(w)
    ops.grad_unprep_parts(ws, slabs, co, gp, "conv", accumulate=True)
    assert rel_err(gp, w.grad) < 2e-3


@pytest.mark.parametrize("case", [(16, 32, 256), (64, 64, 128)])
def test_normalise_on_load_equals_the_materialised_operand(case):
    """Round 3: the streaming 3x3 kernel and the weight-gradient kernel take the RAW conv output of the
    producing layer and apply its InstanceNorm + LeakyReLU in LDS while staging it.  Same bf16 operand values, same MFMA
    order: the results must equal the path through the materialised activation bit for bit (forward, forward with the
    statistics epilogue, 3x3 weight gradient; 32 channels also the 1x1 head's forward and weight gradient)."""
    ops = _ops()
    from cu_hip.engine import TAPS3, TAPS3_W
    n, c, size = case
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(31)
    z = torch.randn(n, size, size, c, device=DEV, generator=g).to(dtype)
    gamma = torch.rand(c, device=DEV, generator=g) + 0.5
    beta = torch.randn(c, device=DEV, generator=g) * 0.3
    raw = ops.instnorm_fwd_fused(z, gamma, beta, 0.01, materialize=False)
    assert raw.a is None and raw.stats is not None
    mat = ops.Act(raw.z, raw.stats, raw.slope)
    ops.instnorm_apply(mat)
    plain = ops.Act(mat.a, None, 1.0)
    w = torch.randn(c, c, 3, 3, device=DEV, generator=g) / math.sqrt(9 * c)
    b = torch.randn(c, device=DEV, generator=g) * 0.1
    wf, _ = ops.weight_prep(w, "conv", dtype)
    outs = []
    for src in (raw, plain):
        o = torch.empty(n, size, size, c, device=DEV, dtype=dtype)
        sums = torch.zeros(n, c, 2, device=DEV)
        got = ops.conv_gemm([src], wf, b, grid=(size, size), in_stride=1, taps=TAPS3, dsts=[o], dst_cols=[c], stat_sums=sums)
        assert got
        o2 = torch.empty_like(o)
        ops.conv_gemm([src], wf, b, grid=(size, size), in_stride=1, taps=TAPS3, dsts=[o2], dst_cols=[c])
        outs.append((o, sums, o2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    assert rel_err(outs[0][1], outs[1][1]) < 1e-5           # f32 atomics: order noise only
    ref = F.conv2d(nchw(mat.a), rq(w, dtype), b, padding=1)
    assert rel_err(nchw(outs[0][0]), ref) < tol(dtype)
    dz = torch.randn(n, size, size, c, device=DEV, generator=g).to(dtype)
    grads = []
    for src in (raw, plain):
        ws = torch.full((24 << 20,), float("nan"), device=DEV)
        slabs = ops.conv_wgrad([src], dz, ws, grid=(size, size), in_stride=1, z_stride=1, taps=TAPS3_W, n_cols=c, parts=True)
        gw = torch.zeros_like(w)
        ops.grad_unprep_parts(ws, slabs, c, gw, "conv", accumulate=True)
        grads.append(gw)
    assert torch.equal(grads[0], grads[1])
    if c == 32:       # the 1x1 head: forward through the generic kernel's fused load, weight gradient through the XF instance
        w1 = torch.randn(21, c, 1, 1, device=DEV, generator=g) / math.sqrt(c)
        wf1, _ = ops.weight_prep(w1, "conv", dtype, cop=32)
        lo = []
        for src in (raw, plain):
            logits = torch.empty(n, 21, size, size, device=DEV)
            ops.conv_gemm([src], wf1, None, grid=(size, size), in_stride=1, taps=[(0, 0, 0)], dsts=[logits], dst_cols=[32],
                          out_nchw=True, n_cols=32)
            lo.append(logits)
        assert rel_err(lo[0], lo[1]) < 1e-5
        dl = torch.randn(n, size, size, 32, device=DEV, generator=g).to(dtype)
        gs = []
        for src in (raw, plain):
            ws = torch.full((8 << 20,), float("nan"), device=DEV)
            slabs = ops.conv_wgrad([src], dl, ws, grid=(size, size), in_stride=1, z_stride=1, taps=[(0, 0, 0, 0, 0)], n_cols=32,
                                   parts=True)
            gw = torch.zeros_like(w1)
            ops.grad_unprep_parts(ws, slabs, 32, gw, "conv", accumulate=True)
            gs.append(gw)
        assert torch.equal(gs[0], gs[1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [
    # n, cin0, cin1, cout, size, stride (output maps of <= 64 pixels at batch 64: the split-K path)
    (64, 480, 0, 480, 4, 1), (64, 480, 480, 480, 4, 1), (64, 480, 0, 480, 2, 1), (64, 480, 0, 480, 4, 2),
    (64, 480, 0, 480, 8, 1), (64, 480, 480, 480, 8, 1), (64, 480, 0, 480, 16, 2), (16, 256, 0, 480, 8, 1),
])
def test_small_map_conv_carries_the_norm_forward(dtype, case):
    """cu_conv_epilogue mode 3: the split-K finish pass writes z, the statistics and LeakyReLU(InstanceNorm(z)).  z must be
    bit-identical to the plain launch; statistics / activation are compared with the separate norm launch and with
    F.instance_norm."""
    ops = _ops()
    from cu_hip.engine import TAPS3
    n, c0, c1, co, size, stride = case
    g = torch.Generator(device=DEV).manual_seed(21)
    srcs = [ops.Act(nhwc(torch.randn(n, c0, size, size, device=DEV, generator=g), dtype), None, 1.0)]
    if c1:
        srcs.append(ops.Act(nhwc(torch.randn(n, c1, size, size, device=DEV, generator=g), dtype), None, 1.0))
    w = torch.randn(co, c0 + c1, 3, 3, device=DEV, generator=g) / math.sqrt(9 * (c0 + c1))
    b = torch.randn(co, device=DEV, generator=g)
    gamma = torch.rand(co, device=DEV, generator=g) + 0.5
    beta = torch.randn(co, device=DEV, generator=g)
    wf, _ = ops.weight_prep(w, "conv", dtype)
    os_ = size // stride
    z0 = torch.empty(n, os_, os_, co, device=DEV, dtype=dtype)
    ops.conv_gemm(srcs, wf, b, grid=(os_, os_), in_stride=stride, taps=TAPS3, dsts=[z0], dst_cols=[co])
    old = ops.instnorm_fwd_fused(z0, gamma, beta, 0.01, 1e-5)
    z = torch.full_like(z0, float("nan"))
    stats = torch.full((4, n, co), float("nan"), device=DEV)
    a = torch.full_like(z0, float("nan"))
    got = ops.conv_gemm(srcs, wf, b, grid=(os_, os_), in_stride=stride, taps=TAPS3, dsts=[z], dst_cols=[co],
                        norm_fwd=(gamma, beta, 1e-5, 0.01, stats, a))
    if not got:
        assert n < 64                       # too much work per tile for a split: the plain result must be there
        assert torch.equal(z, z0)
        return
    # (a split the plain launch did not make -- 8x8 -- sums in another order: not bit-identical there)
    assert torch.equal(z, z0) or (os_ == 8 and rel_err(z.float(), z0.float()) < (1e-5 if dtype == torch.float32 else 8e-3))
    zs = z.float()
    ref = F.leaky_relu(F.instance_norm(nchw(z).float(), weight=gamma, bias=beta, eps=1e-5), 0.01)
    assert rel_err(nchw(a), ref) < tol(dtype)
    mean = zs.mean((1, 2))
    rstd = (zs.var((1, 2), unbiased=False) + 1e-5).rsqrt()
    assert rel_err(stats[0], mean) < 1e-5 and rel_err(stats[1], rstd) < 1e-4
    assert rel_err(stats[2], gamma * rstd) < 1e-4 and rel_err(stats[3], beta - mean * gamma * rstd) < 1e-4
    if torch.equal(z, z0):
        assert rel_err(stats, old.stats) < 1e-4
        assert rel_err(a.float(), old.a.float()) < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["s1", "s2", "s2_accum", "convT", "convT_accum"])
@pytest.mark.parametrize("size", [2, 4, 8])
def test_small_map_input_gradient_carries_the_norm_backward(dtype, kind, size):
    """cu_conv_epilogue mode 4: the split-K finish pass of an input-gradient launch writes dL/dz of the layer whose
    activation it differentiates (+ dgamma, dbeta) -- against the plain launch followed by the norm-backward launch."""
    ops = _ops()
    from cu_hip.engine import TAPS3_D, S2_PARITY_TAPS
    n, c = 64, 480
    g = torch.Generator(device=DEV).manual_seed(31 + size)
    # the target layer: z, statistics, activation on a size x size map
    zt = nhwc(torch.randn(n, c, size, size, device=DEV, generator=g) * 2 + 0.3, dtype)
    gamma = torch.rand(c, device=DEV, generator=g) + 0.5
    beta = torch.randn(c, device=DEV, generator=g) * 0.2
    tgt = ops.instnorm_fwd_fused(zt, gamma, beta, 0.01, 1e-5)
    accum = kind.endswith("accum")
    base = nhwc(torch.randn(n, c, size, size, device=DEV, generator=g), dtype)
    w = torch.randn(c, c, 3, 3, device=DEV, generator=g) / math.sqrt(9 * c)
    if kind == "s1":
        dz = ops.Act(nhwc(torch.randn(n, c, size, size, device=DEV, generator=g), dtype), None, 1.0)
        _, wd = ops.weight_prep(w, "conv", dtype)
        kw = dict(grid=(size, size), in_stride=1, taps=TAPS3_D, dst_cols=[c], accum=[0])
    elif kind.startswith("s2"):
        if size < 4 or dtype != torch.bfloat16:
            pytest.skip("one-pass stride-2 input gradient: bf16, destination >= 4x4")
        os_ = size // 2
        dz = ops.Act(nhwc(torch.randn(n, c, os_, os_, device=DEV, generator=g), dtype), None, 1.0)
        _, wd = ops.weight_prep(w, "conv", dtype)
        kw = dict(grid=(os_, os_), in_stride=1, taps=[(u, v, 0) for u in range(2) for v in range(2)], dst_cols=[c],
                  out_stride=2, accum=[int(accum)], n_cols=4 * c, parity_cols=c, parity_taps=S2_PARITY_TAPS)
    else:
        wt = torch.randn(c, c, 2, 2, device=DEV, generator=g) / math.sqrt(4 * c)
        dz = ops.Act(nhwc(torch.randn(n, c, 2 * size, 2 * size, device=DEV, generator=g), dtype), None, 1.0)
        _, wd = ops.weight_prep(wt, "convT", dtype)
        kw = dict(grid=(size, size), in_stride=2, taps=[(dy, dx, dy * 2 + dx) for dy in range(2) for dx in range(2)],
                  dst_cols=[c], accum=[int(accum)])
    # plain launch + norm backward launch
    d_ref = base.clone()
    ops.conv_gemm([dz], wd, None, dsts=[d_ref], **kw)
    dg_ref, db_ref = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    ops.instnorm_bwd_fused(d_ref, tgt, gamma, dg_ref, db_ref)
    # fused
    d = base.clone()
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    got = ops.conv_gemm([dz], wd, None, dsts=[d], norm_bwd_full=(tgt, gamma, dg, db), **kw)
    assert got, "the launch did not take the fused finish"
    t = 2e-5 if dtype == torch.float32 else 1.2e-2
    assert rel_err(d.float(), d_ref.float()) < t
    # (bf16 at 8x8: the plain launch does not split there, g rounds from another summation order)
    tg = 1e-4 if dtype == torch.float32 else 1e-3
    assert rel_err(dg, dg_ref) < tg and rel_err(db, db_ref) < tg


@pytest.mark.parametrize("case", [
    # n, cin0, cin1, cout, size, stride
    (16, 128, 0, 128, 64, 1),        # ring kernel, one image per tile
    (16, 128, 128, 128, 64, 1),      # two sources
    (64, 256, 0, 256, 32, 1),
    (64, 480, 0, 480, 16, 1),        # two images per tile, ragged last column tile
    (63, 480, 0, 480, 16, 1),        # odd batch: the last tile's second image does not exist
    (16, 64, 0, 128, 128, 2),        # stride-2 gather
    (64, 256, 0, 480, 32, 2),
])
def test_wide_conv_gathers_the_norm_statistics(case):
    """cu_conv_epilogue mode 1 on the LDS-DMA kernels: sums [N][CO][2] += {sum, sum of squares} of (output - bias) over
    each image, from the f32 accumulators -- against the same sums of an f32 convolution of the same bf16 operands."""
    ops = _ops()
    from cu_hip.engine import TAPS3
    dtype = torch.bfloat16
    n, c0, c1, co, size, stride = case
    g = torch.Generator(device=DEV).manual_seed(41)
    xs = [rq(torch.randn(n, c0, size, size, device=DEV, generator=g), dtype)]
    if c1:
        xs.append(rq(torch.randn(n, c1, size, size, device=DEV, generator=g), dtype))
    srcs = [ops.Act(nhwc(x, dtype), None, 1.0) for x in xs]
    w = torch.randn(co, c0 + c1, 3, 3, device=DEV, generator=g) / math.sqrt(9 * (c0 + c1))
    b = torch.randn(co, device=DEV, generator=g)
    wf, _ = ops.weight_prep(w, "conv", dtype)
    os_ = size // stride
    z0 = torch.empty(n, os_, os_, co, device=DEV, dtype=dtype)
    ops.conv_gemm(srcs, wf, b, grid=(os_, os_), in_stride=stride, taps=TAPS3, dsts=[z0], dst_cols=[co])
    z = torch.empty_like(z0)
    sums = torch.zeros(n, co, 2, device=DEV)
    got = ops.conv_gemm(srcs, wf, b, grid=(os_, os_), in_stride=stride, taps=TAPS3, dsts=[z], dst_cols=[co], stat_sums=sums)
    assert got, "the launch did not gather the statistics"
    assert torch.equal(z, z0)
    ref = F.conv2d(torch.cat(xs, 1), rq(w, dtype), None, stride=stride, padding=1)
    s1, s2 = ref.sum((2, 3)), (ref * ref).sum((2, 3))
    assert rel_err(sums[:, :, 0], s1) < 2e-4          # (sums near zero against the largest: f32 summation order)
    assert rel_err(sums[:, :, 1], s2) < 2e-5
    gamma = torch.rand(co, device=DEV, generator=g) + 0.5
    beta = torch.randn(co, device=DEV, generator=g)
    out = ops.instnorm_fwd_given(z, gamma, beta, 0.01, sums, b)
    refa = F.leaky_relu(F.instance_norm(nchw(z).float(), weight=gamma, bias=beta, eps=1e-5), 0.01)
    assert rel_err(nchw(out.a), refa) < tol(dtype)


@pytest.mark.parametrize("size,c,n", [(16, 480, 6), (32, 256, 5), (16, 64, 3), (32, 96, 2)])
@pytest.mark.parametrize("parts", [False, True])
def test_instnorm_bwd_small_resident_form_equals_the_default(size, c, n, parts):
    """CU_NORM_SMALL_RES (register-resident 16x16 / 32x32 backward, bf16; opt-in since round 4: faster alone, slower inside the
    step) against the default two-pass small-map kernel: same dz, same dgamma / dbeta -- summed over the images by atomics or
    left as per-image planes for cu_norm_param_grads_batch (CU_NORM_PARAM_PARTS)."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(31)
    z = torch.randn(n, size, size, c, device=DEV, generator=g).to(torch.bfloat16)
    gin = torch.randn(n, size, size, c, device=DEV, generator=g).to(torch.bfloat16)
    gamma = torch.rand(c, device=DEV, generator=g) + 0.5
    beta = torch.randn(c, device=DEV, generator=g) * 0.2
    act = ops.instnorm_fwd_fused(z, gamma, beta, 0.01, 1e-5)
    res = []
    for flag in (0, ops.NORM_SMALL_RES):
        dz = gin.clone()
        if parts:
            planes = torch.zeros(2, n, c, device=DEV)
            ops.instnorm_bwd_fused(dz, act, gamma, planes[0], planes[1], mode=2 | ops.NORM_PARAM_PARTS | flag)
            dg, db = planes[0].sum(0), planes[1].sum(0)
        else:
            dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
            ops.instnorm_bwd_fused(dz, act, gamma, dg, db, mode=2 | flag)
        res.append((dz.float(), dg, db))
    torch.cuda.synchronize()
    (dz0, dg0, db0), (dz1, dg1, db1) = res
    assert rel_err(dz1, dz0) < 8e-3                       # both round dz to bf16 once; the sums differ in the last f32 bits
    assert rel_err(dg1, dg0) < 1e-5 and rel_err(db1, db0) < 1e-5
