"""SURVEY.md section 5 (sanitizers) / VERDICT r2 item 8: the HOST half of libcontour_hip.so -- argument validation,
descriptor decoding, launch-wrapper arithmetic, comm.cpp, error.cpp -- under AddressSanitizer.  CPU only: the device code of
the `make ASAN=1` build is unsanitised (GPU ASan / xnack+ are not available on this pool) and no kernel is launched: every
call below must be refused by its wrapper with an errno-style code before it touches HIP."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "contouring-uncertainty_amd"
RUNTIME = Path("/opt/rocm/lib/llvm/lib/clang")

SCRIPT = r'''
import ctypes as C, sys
sys.path.insert(0, r"%(pkg)s")
from cu_hip import lib
h = lib.load()
P = C.c_void_p
# null descriptors / pointers, bad dtypes, non-square maps, bad layouts, empty workspaces: all -22 with a message
assert h.cu_conv_gemm(None, *([None] * 11)) == -22 and b"null descriptor" in h.cu_last_error()
d = lib.ConvDesc(); d.dtype = 7
assert h.cu_conv_gemm(d, *([None] * 11)) == -22
d = lib.ConvDesc(); d.dtype = 1; d.ntaps = 99
assert h.cu_conv_gemm(d, *([None] * 11)) == -22
assert h.cu_conv_gemm_ws(None, *([None] * 11), 0, None) == -22
ep = lib.ConvEpilogue(5, None, None, None, 1.0)
assert h.cu_conv_gemm_ex(None, *([None] * 11), 0, C.byref(ep), None, None) == -22
w = lib.WgradDesc()
assert h.cu_conv_wgrad(None, *([None] * 9)) == -22
w.dtype = 1; w.ntaps = 0
assert h.cu_conv_wgrad(w, *([None] * 9)) == -22
w.ntaps = 9; w.C0 = 33
assert h.cu_conv_wgrad(w, *([None] * 9)) == -22          # channel count not a multiple of 8
n, l = C.c_int(0), C.c_int(0)
assert h.cu_conv_wgrad_parts(w, *([None] * 8), 0, C.byref(n), C.byref(l), None) == -22
assert h.cu_conv_wgrad_parts(w, *([None] * 8), 0, None, None, None) == -22
assert h.cu_grad_unprep_parts(9, 64, 64, 64, 576, 9, None, 0, 1, 0x202, None, 1, None) == -22
assert h.cu_grad_unprep(0, 0, 0, 0, 0, 0, None, None, 0, None) == -22
assert h.cu_dsnt_head_fwd(4, 16, 24, None, 1, None, None, None, None) == -22 and b"square" in h.cu_last_error()
assert h.cu_weight_prep(3, 9, 8, 8, 8, 72, 9, None, None, None, None) == -22
assert h.cu_adam_step(0, None, None, None, None, 1e-3, .9, .999, 1e-8, 0., 1, 1., None) == -22
assert h.cu_adam_step_dev(0, None, None, None, None, 1e-3, .9, .999, 1e-8, 0., None, 1., None) == -22
assert h.cu_comm_init(0, 0, None, None) == -22
assert h.cu_comm_allreduce_bucket(None, None, 0, None) == -22
assert h.cu_comm_destroy(None) in (0, -22)
assert h.cu_version() >= 100 and h.cu_arch() == b"gfx950"
# the error text buffer is bounded (a 600-character layer name must not overrun it)
for _ in range(3):
    h.cu_dsnt_head_fwd(4, 16, 24, None, 1, None, None, None, None)
assert len(h.cu_last_error()) < 512
print("asan-host-ok")
'''


def _runtime():
    hits = sorted(RUNTIME.glob("*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


@pytest.mark.timeout(900)
def test_host_side_under_address_sanitizer():
    rt = _runtime()
    if rt is None:
        pytest.skip("clang ASan runtime not in this image")
    lib = PKG / "libcontour_hip_asan.so"
    subprocess.run(["make", "-C", str(PKG / "csrc"), "ASAN=1", "-j8"], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)      # incremental: seconds when up to date, ~2 min from scratch
    env = dict(os.environ, LD_PRELOAD=str(rt), CONTOUR_HIP_LIB=str(lib),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66:protect_shadow_gap=0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"pkg": str(PKG)}], env=env, capture_output=True, text=True)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "asan-host-ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
