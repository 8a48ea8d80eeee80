"""The product's seeded initialisation (what ``bench.py``'s parity field loads) draws exactly the tensors the golden
vectors were made with (``oracle.unet.init_unet_state`` / ``init_confidence_state`` with the same generator)."""
import torch

from oracle import unet as OU


def test_seeded_state_equals_the_oracle_initialisation():
    from contour_uncertainty.data.synthetic.weights import seeded_confidence_state, seeded_unet_state
    from contour_uncertainty.models.nnUnet.unet2 import ConfidenceNet, UNet
    spec = OU.UNetSpec(strides=(1, 2, 2, 2, 2, 2))
    net = UNet((1, 64, 64), (21, 1, 64), [256, 256], [[3, 3]] * 6, [[1, 1]] + [[2, 2]] * 5, bottleneck_out=True,
               compute_dtype="f32")
    head = ConfidenceNet(42, compute_dtype="f32")
    g0, g1 = torch.Generator().manual_seed(0), torch.Generator().manual_seed(0)
    ref, ref_h = OU.init_unet_state(spec, g0), OU.init_confidence_state(42, g0)
    got, got_h = seeded_unet_state(net, g1), seeded_confidence_state(head, g1)
    assert list(ref) == list(got) and list(ref_h) == list(got_h)
    assert all(torch.equal(ref[k], got[k]) for k in ref) and all(torch.equal(ref_h[k], got_h[k]) for k in ref_h)
