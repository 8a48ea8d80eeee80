import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "contouring-uncertainty_amd"))
import numpy as np, torch, torch.nn.functional as F
from oracle import vital_unet as OV
from cu_hip import ops
from cu_hip.engine_vital import VitalUNetEngine
g = np.load(ROOT / "tests/golden/vital_unet.npz")
sd = OV.init_state(1, 5, 32, torch.Generator().manual_seed(23))
x = torch.from_numpy(g["x"])
P = {k: v.cuda() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
S = {k: v.clone().cuda() for k, v in sd.items() if k not in P}
eng = VitalUNetEngine(1, 5, 32, torch.float32)
def nchw(t): return t.permute(0, 3, 1, 2).float().cpu()
# stage 1: first conv
z_ref = F.conv2d(x, sd["layer1.net.0.weight"], sd["layer1.net.0.bias"], padding=1)
w9, _ = eng._first_operand("a", P["layer1.net.0.weight"])
z = torch.empty(3, 64, 64, 16, device="cuda")
ops.conv_c1_fwd(x.cuda(), w9, P["layer1.net.0.bias"], z)
print("conv_c1", float((nchw(z) - z_ref).abs().max()), float(z_ref.abs().max()))
a_ref = F.relu(F.batch_norm(z_ref, None, None, sd["layer1.net.1.weight"], sd["layer1.net.1.bias"], True, 0.1, 1e-5))
act = eng._bn_relu(P, S, "layer1.net.1", z, True)
a = act.a.view(3, 64, 64, 16)
print("bn_relu", float((nchw(a) - a_ref).abs().max()), float(a_ref.abs().max()))
z2_ref = F.conv2d(a_ref, sd["layer1.net.4.weight"], sd["layer1.net.4.bias"], padding=1)
wf, _ = eng._operands("b", P["layer1.net.4.weight"], "conv")
z2 = torch.empty(3, 64, 64, 16, device="cuda")
from cu_hip.engine import TAPS3
ops.conv_gemm([ops.Act(a, None, 1.0)], wf, P["layer1.net.4.bias"], grid=(64, 64), in_stride=1, taps=TAPS3, dsts=[z2], dst_cols=[16])
print("conv 16->16", float((nchw(z2) - z2_ref).abs().max()), float(z2_ref.abs().max()))
