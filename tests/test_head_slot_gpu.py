"""The dL/dlogits hand-over between the two autograd nodes (cu_hip.head.GradSlot): two loss terms on ONE logits
tensor must both reach the network's parameters, and the dense (no-slot) path must agree with the hand-over."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(dtype):
    from contour_uncertainty.models.nnUnet.unet2 import UNet
    torch.manual_seed(5)
    net = UNet((1, 32, 32), (5, 1, 32), [256, 256], [[3, 3]] * 4, [[1, 1]] + [[2, 2]] * 3, compute_dtype=dtype)
    return net.to(DEV)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_heads_on_one_logits_accumulate(dtype):
    from cu_hip.head import dsnt_nll
    net = _net(dtype)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand(2, 1, 32, 32, device=DEV, generator=g)
    y1 = torch.rand(2, 5, 2, device=DEV, generator=g) * 31
    y2 = torch.rand(2, 5, 2, device=DEV, generator=g) * 31

    def grads(dense):
        net.zero_grad(set_to_none=True)
        logits = net(x)
        l1 = dsnt_nll(logits, y1, None, True, dense_grad=dense)[0]["loss"]
        l2 = dsnt_nll(logits, y2, None, False, dense_grad=dense)[0]["loss"]
        (l1 + 0.5 * l2).backward()
        return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    slot, dense = grads(False), grads(True)
    assert slot.keys() == dense.keys()
    tol = 2e-4 if dtype == "f32" else 6e-2     # bf16: the hand-over rounds the SUM once, the dense path converts it once too
    for n in slot:
        if n.endswith("conv.bias") and "output" not in n:
            continue
        d = float((slot[n] - dense[n]).norm() / dense[n].norm().clamp_min(1e-12))
        assert d < tol, (n, d)
    # and a single head is NOT the same gradient (the second term really arrives)
    net.zero_grad(set_to_none=True)
    dsnt_nll(net(x), y1, None, True)[0]["loss"].backward()
    one = net.output_block.conv.weight.grad
    assert float((one - slot["output_block.conv.weight"]).norm() / one.norm()) > 1e-2
