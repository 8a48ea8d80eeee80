"""Measured error of the f32 parity mode at full size (8 stages, 256x256, N = 1) against the reference golden
(tests/golden/unet_full.npz): what tests/test_model_gpu.py::test_full_size_forward_vs_reference_golden bounds."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import numpy as np
import torch
from oracle import unet as OU      # tool, not product: the seeded state the golden was made with
from contour_uncertainty.models.nnUnet.unet2 import ConfidenceNet, UNet

g = np.load(ROOT / "tests" / "golden" / "unet_full.npz")
spec = OU.UNetSpec()
gen = torch.Generator().manual_seed(0)
sd = OU.init_unet_state(spec, gen)
ssd = OU.init_confidence_state(42, gen)
x = torch.rand(1, 1, 256, 256, generator=gen)
net = UNet((1, 256, 256), (21, 1, 256), [256, 256], [[3, 3]] * 8, [[1, 1]] + [[2, 2]] * 7, bottleneck_out=True, compute_dtype="f32")
net.load_state_dict(sd, strict=True)
head = ConfidenceNet(42, compute_dtype="f32")
head.load_state_dict(ssd, strict=True)
net, head = net.cuda(), head.cuda()
with torch.no_grad():
    logits, bott = net(x.cuda())
    a = head(bott)
for name, got, ref in (("bottleneck", bott.cpu(), g["bottleneck"]), ("alpha_raw", a.cpu(), g["alpha_raw"]),
                       ("logits_row", logits[0, :, 128, :].cpu(), g["logits_row"])):
    ref = torch.from_numpy(ref)
    d = (got.double() - ref.double()).abs()
    print(f"{name:12s} max abs {float(d.max()):.3e}  max |ref| {float(ref.abs().max()):.3e}  "
          f"allclose(3e-4, 1e-4) {torch.allclose(got, ref, rtol=3e-4, atol=1e-4)}  (1e-3, 3e-4) {torch.allclose(got, ref, rtol=1e-3, atol=3e-4)}")
