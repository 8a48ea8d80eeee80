"""ctypes binding of libcontour_hip.so (the C ABI declared in include/contour_hip.h).

The product path has NO fallback: if the shared library is missing, or a kernel is asked to run without a GPU,
an exception is raised.  PyTorch is used only for device memory and streams (tensors' ``data_ptr()``, the current
HIP stream handle).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

PKG_DIR = Path(__file__).resolve().parents[1]
LIB_PATH = PKG_DIR / "libcontour_hip.so"

CU_F32, CU_BF16 = 0, 1
MAX_TAPS = 9

_INT9 = C.c_int * MAX_TAPS


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("N", C.c_int), ("PH", C.c_int), ("PW", C.c_int), ("SH", C.c_int), ("SW", C.c_int),
        ("C0", C.c_int), ("C1", C.c_int), ("IS", C.c_int), ("OH", C.c_int), ("OW", C.c_int), ("OS", C.c_int),
        ("OY0", C.c_int), ("OX0", C.c_int), ("CO", C.c_int), ("D0", C.c_int), ("DC0", C.c_int), ("DC1", C.c_int),
        ("ntaps", C.c_int), ("tap_dy", _INT9), ("tap_dx", _INT9), ("tap_w", _INT9),
        ("slope0", C.c_float), ("slope1", C.c_float), ("accum0", C.c_int), ("accum1", C.c_int),
        ("out_nchw_f32", C.c_int), ("par_co", C.c_int), ("par_taps", C.c_int), ("par_tap_w", C.c_int * 16),
    ]


class ConvEpilogue(C.Structure):      # cu_conv_epilogue
    _fields_ = [("mode", C.c_int), ("sums", C.c_void_p), ("z", C.c_void_p), ("stats", C.c_void_p), ("slope", C.c_float),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("act_out", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("N", C.c_int), ("PH", C.c_int), ("PW", C.c_int), ("SH", C.c_int), ("SW", C.c_int),
        ("C0", C.c_int), ("C1", C.c_int), ("IS", C.c_int), ("ZH", C.c_int), ("ZW", C.c_int), ("ZC", C.c_int),
        ("ZS", C.c_int), ("CO", C.c_int), ("ntaps", C.c_int), ("tap_dy", _INT9), ("tap_dx", _INT9),
        ("tap_zy", _INT9), ("tap_zx", _INT9), ("tap_w", _INT9), ("slope0", C.c_float), ("slope1", C.c_float),
        ("splits", C.c_int),
    ]


_P = C.c_void_p
_SIGS = {
    "cu_last_error": (C.c_char_p, []),
    "cu_version": (C.c_int, []),
    "cu_arch": (C.c_char_p, []),
    "cu_conv_gemm": (C.c_int, [C.POINTER(ConvDesc)] + [_P] * 11),
    "cu_conv_gemm_ws": (C.c_int, [C.POINTER(ConvDesc)] + [_P] * 11 + [C.c_size_t, _P]),
    "cu_conv_gemm_stats": (C.c_int, [C.POINTER(ConvDesc)] + [_P] * 11 + [C.c_size_t, _P, C.POINTER(C.c_int), _P]),
    "cu_conv_gemm_ex": (C.c_int, [C.POINTER(ConvDesc)] + [_P] * 11 + [C.c_size_t, C.POINTER(ConvEpilogue), C.POINTER(C.c_int), _P]),
    "cu_conv_wgrad": (C.c_int, [C.POINTER(WgradDesc)] + [_P] * 9),
    "cu_conv_wgrad_parts": (C.c_int, [C.POINTER(WgradDesc)] + [_P] * 8 + [C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    "cu_conv_c1_fwd": (C.c_int, [C.c_int] * 5 + [_P] * 5),
    "cu_conv_c1_norm_ws_floats": (C.c_size_t, [C.c_int] * 3),
    "cu_conv_c1_fwd_norm": (C.c_int, [C.c_int] * 5 + [_P] * 5 + [C.c_float] * 2 + [_P] * 5),
    "cu_conv_c1_bwd": (C.c_int, [C.c_int] * 5 + [_P] * 5 + [C.c_float] + [_P] * 6),
    "cu_conv_c1_wgrad": (C.c_int, [C.c_int] * 5 + [_P] * 4),
    "cu_conv_c1_wgrad_det": (C.c_int, [C.c_int] * 5 + [_P] * 4 + [C.c_size_t, _P]),
    "cu_instnorm_stats": (C.c_int, [C.c_int] * 4 + [_P] * 3 + [C.c_float] + [_P] * 3),
    "cu_instnorm_apply": (C.c_int, [C.c_int] * 4 + [_P] * 2 + [C.c_float] + [_P] * 2),
    "cu_instnorm_lrelu_bwd": (C.c_int, [C.c_int] * 4 + [_P] * 4 + [C.c_float] + [_P] * 5),
    "cu_instnorm_resident_ws_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "cu_instnorm_fwd_fused": (C.c_int, [C.c_int] * 4 + [_P] * 3 + [C.c_float, C.c_float] + [_P] * 3 + [C.c_int, _P]),
    "cu_instnorm_fwd_given": (C.c_int, [C.c_int] * 4 + [_P] * 3 + [C.c_float, C.c_float] + [_P] * 5),
    "cu_instnorm_bwd_given": (C.c_int, [C.c_int] * 4 + [_P] * 4 + [C.c_float] + [_P] * 4),
    "cu_instnorm_bwd_fused": (C.c_int, [C.c_int] * 4 + [_P] * 4 + [C.c_float] + [_P] * 3 + [C.c_int, _P]),
    "cu_norm_param_grads_batch": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "cu_channel_scale": (C.c_int, [C.c_int] * 4 + [_P] * 3),
    "cu_maxpool2_fwd": (C.c_int, [C.c_int] * 5 + [_P] * 4),
    "cu_maxpool2_bwd": (C.c_int, [C.c_int] * 5 + [_P] * 4),
    "cu_act_bwd": (C.c_int, [C.c_int] * 4 + [_P] * 2 + [C.c_float] + [_P] * 2),
    "cu_act_bwd_det": (C.c_int, [C.c_int] * 4 + [_P] * 2 + [C.c_float] + [_P] * 2),
    "cu_act_to_nchw_f32": (C.c_int, [C.c_int] * 4 + [_P] * 2 + [C.c_float] + [_P] * 2),
    "cu_nchw_f32_to_nhwc": (C.c_int, [C.c_int] * 5 + [_P] * 3),
    "cu_nhwc_to_nchw_f32": (C.c_int, [C.c_int] * 4 + [_P] * 2 + [C.c_int, _P]),
    "cu_dsnt_head_fwd": (C.c_int, [C.c_int] * 3 + [_P, C.c_int] + [_P] * 4),
    "cu_dsnt_head_bwd": (C.c_int, [C.c_int] * 3 + [_P] * 4 + [C.c_int] + [_P] * 2),
    "cu_dsnt_head_bwd_nhwc": (C.c_int, [C.c_int] * 5 + [_P] * 4 + [C.c_int] + [_P] * 2),
    "cu_head_fused_ws_floats": (C.c_size_t, [C.c_int] * 3),
    "cu_head_fused_fwd": (C.c_int, [C.c_int] * 4 + [_P, _P, C.c_float, _P, C.c_int, _P, C.c_size_t] + [_P] * 4),
    "cu_head_fused_bwd": (C.c_int, [C.c_int] * 4 + [_P, _P, C.c_float] + [_P] * 5 + [C.c_int] + [_P] * 3 + [C.c_size_t, C.POINTER(C.c_int), _P]),
    "cu_nll_fwd_bwd": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float] + [_P] * 10),
    "cu_linear_fwd": (C.c_int, [C.c_int] * 3 + [_P] * 5),
    "cu_linear_bwd": (C.c_int, [C.c_int] * 3 + [_P] * 7),
    "cu_weight_prep": (C.c_int, [C.c_int] * 5 + [C.c_long, C.c_long] + [_P] * 4),
    "cu_grad_unprep": (C.c_int, [C.c_int] * 4 + [C.c_long, C.c_long] + [_P] * 2 + [C.c_int, _P]),
    "cu_grad_unprep_parts": (C.c_int, [C.c_int] * 4 + [C.c_long, C.c_long, _P, C.c_size_t, C.c_int, C.c_int, _P, C.c_int, _P]),
    "cu_psm_sample_gauss": (C.c_int, [C.c_int] * 3 + [_P] * 6 + [C.c_int, _P, C.c_int] + [_P] * 4 + [C.c_uint64, _P, _P]),
    "cu_weight_prep_batch": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P]),
    "cu_psm_record_floats": (C.c_int, [C.c_int, _P, _P]),
    "cu_psm_setup": (C.c_int, [C.c_int] * 2 + [_P] * 5 + [C.c_int, _P, _P, _P, C.c_int, _P]),
    "cu_psm_sample_skew": (C.c_int, [C.c_int] * 3 + [_P] * 3 + [C.c_float, C.c_uint64, _P, C.c_int, _P, _P, C.c_int, _P,
                                     C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_uint64, _P, _P]),
    "cu_psm_condition": (C.c_int, [C.c_int] * 3 + [_P, C.c_int, _P, C.c_int] + [_P] * 10),
    "cu_contour_masks": (C.c_int, [C.c_int] * 4 + [_P, C.c_int, C.c_int, _P, _P, _P]),
    "cu_contour_measures": (C.c_int, [C.c_int] * 4 + [_P, C.c_int, _P, _P, _P]),
    "cu_mask_last_value": (C.c_int, [C.c_int] * 3 + [_P] * 4),
    "cu_mask_entropy": (C.c_int, [C.c_int] * 4 + [_P] * 4),
    "cu_mask_weighted_entropy": (C.c_int, [C.c_int] * 4 + [_P] * 5),
    "cu_logpdf_grid": (C.c_int, [C.c_int] * 3 + [_P] * 6),
    "cu_skew_rvs": (C.c_int, [C.c_int] * 2 + [_P] * 4 + [C.c_uint64, _P, _P]),
    "cu_augment_image": (C.c_int, [C.c_int] * 3 + [_P] * 5),
    "cu_augment_labels": (C.c_int, [C.c_int] * 3 + [_P] * 4),
    "cu_comm_unique_id": (C.c_int, [_P]),
    "cu_comm_init": (C.c_int, [C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "cu_comm_allreduce_bucket": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cu_comm_reduce_scatter_bucket": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "cu_comm_allgather_bucket": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "cu_comm_destroy": (C.c_int, [_P]),
    "cu_adam_step": (C.c_int, [C.c_size_t] + [_P] * 4 + [C.c_float] * 5 + [C.c_int, C.c_float, _P]),
    "cu_adam_step_dev": (C.c_int, [C.c_size_t] + [_P] * 4 + [C.c_float] * 5 + [_P, C.c_float, _P]),
    "cu_step_advance": (C.c_int, [_P, _P]),
}

_lib = None


class ContourHipError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for loading); raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("CONTOUR_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise ContourHipError(
            f"{path} not found: build it with `make -C {PKG_DIR / 'csrc'}` (or __graft_entry__.build()). "
            "There is no CPU/PyTorch fallback for the HIP kernels.")
    lib = C.CDLL(str(path))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGS.keys())


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().cu_last_error().decode()
        raise ContourHipError(f"{what} failed ({rc}): {msg}")


_first_operand = [None]      # first tensor handed to ptr() since the last stream_ptr(): one device check per launch


def stream_ptr() -> int:
    """The current HIP stream of the CURRENT device: kernels launch on the thread's current device, so the operands
    must live there (checked here on the first operand of the call -- the wrappers allocate every output on that
    operand's device; ``device_guard`` makes an operand's device current)."""
    t, _first_operand[0] = _first_operand[0], None
    dev = _get_device()
    if t is not None and t.device.index != dev:
        raise ContourHipError(f"operand on {t.device} but the current device is cuda:{dev}: "
                              "launches go to the current device's stream (use torch.cuda.set_device / "
                              "cu_hip.lib.device_guard)")
    # (the raw handle straight from the C side: torch.cuda.current_stream() builds a Stream object per call, ~4 us of the
    # host's ~25 us per launch -- tools/cpu_bound.py)
    return _raw_stream(dev)


def _get_device() -> int:
    f = getattr(torch._C, "_cuda_getDevice", None)
    return f() if f is not None else torch.cuda.current_device()


def _raw_stream(dev: int) -> int:
    f = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    return f(dev) if f is not None else torch.cuda.current_stream().cuda_stream


def device_guard(t):
    """Context manager that makes ``t``'s device the current one for the launches inside (a module moved with
    ``.to('cuda:1')`` while cuda:0 is current would otherwise enqueue its kernels on the wrong GPU and stream)."""
    return torch.cuda.device(t.device)


def ptr(t):
    """data_ptr of a CUDA tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ContourHipError("HIP kernels need device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise ContourHipError("HIP kernels need contiguous tensors")
    if _first_operand[0] is None:
        _first_operand[0] = t
    return t.data_ptr()


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return CU_F32
    if dt == torch.bfloat16:
        return CU_BF16
    raise ContourHipError(f"unsupported element type {dt}")


def require_gpu():
    if not torch.cuda.is_available():
        raise ContourHipError("no MI355X visible: the contour HIP path has no CPU fallback")
