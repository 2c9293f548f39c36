"""ctypes binding of libfcd_hip.so (include/fcd_hip.h).

The product path has no fallback: if the shared library is missing, or a call
returns a non-zero status, a RuntimeError is raised.  PyTorch is used here only
for device memory (``tensor.data_ptr()``), the current HIP stream and the
caching allocator that provides workspaces.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FCD_HIP_LIB: another build of the same library (A/B timing of kernel variants in one GPU session)
LIB_PATH = os.environ.get("FCD_HIP_LIB") or os.path.join(_HERE, "libfcd_hip.so")

ABI_VERSION = 2    # include/fcd_hip.h: FCD_ABI_VERSION
PREC_F32 = 0
PREC_BF16 = 1
PREC_BF16X3 = 2
PREC_F16X2 = 3     # fp16 x 2 pieces on the 3x3 A-halo layers, BF16X3 elsewhere
PREC_F16 = 4       # FCD_PREC_F16: the same kernels and operands, high fp16 piece only (one MFMA per product group)
F16_FAMILY = (PREC_F16X2, PREC_F16)   # precisions whose convolutions take the scaled fp16 operands (maxima attached)

c_float_p = C.c_void_p  # device pointers travel as opaque addresses


class ConvDesc(C.Structure):
    """Mirror of ``FcdConvDesc``."""
    _fields_ = [
        ("N", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32), ("Cin", C.c_int32),
        ("GH", C.c_int32), ("GW", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32), ("CoutP", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32),
        ("istride", C.c_int32), ("ostride", C.c_int32),
        ("nphase", C.c_int32),
        ("pad_h", C.c_int32 * 4), ("pad_w", C.c_int32 * 4),
        ("ooff_h", C.c_int32 * 4), ("ooff_w", C.c_int32 * 4),
        ("splitk", C.c_int32), ("precision", C.c_int32),
        ("act_slope", C.c_float), ("mask_slope", C.c_float),
        ("a_amax", C.c_void_p), ("w_amax", C.c_void_p),          # FCD_PREC_F16X2 only
        ("out_amax", C.c_void_p),                                # optional: max |out| of the launch (AMAX_SLOTS floats)
    ]


class WgradDesc(C.Structure):
    """Mirror of ``FcdWgradDesc``."""
    _fields_ = [
        ("N", C.c_int32), ("GH", C.c_int32), ("GW", C.c_int32), ("O", C.c_int32), ("OP", C.c_int32),
        ("BH", C.c_int32), ("BW", C.c_int32), ("I", C.c_int32), ("IP", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("splitk", C.c_int32), ("precision", C.c_int32), ("accumulate", C.c_int32),
        ("g_amax", C.c_void_p), ("x_amax", C.c_void_p),          # FCD_PREC_F16X2 only
    ]


_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/fcd_hip.h
PROTOTYPES = {
    "fcd_last_error": (C.c_char_p, []),
    "fcd_abi_version": (_I, []),
    "fcd_init": (_I, [_I]),
    "fcd_conv_desc_bytes": (_I, []),
    "fcd_wgrad_desc_bytes": (_I, []),
    "fcd_set_overflow_flag": (_I, [_I, _P]),
    "fcd_last_conv_kernel": (C.c_char_p, []),
    "fcd_conv_workspace_bytes": (_Z, [C.POINTER(ConvDesc)]),
    "fcd_conv_gather": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "fcd_conv_gather_stats_workspace_bytes": (_Z, [C.POINTER(ConvDesc)]),
    "fcd_conv_gather_stats": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _F, _P, _Z, _P]),
    "fcd_conv_gather_stats_apply": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _F, _P, _P, _Z, _P]),
    "fcd_wgrad_workspace_bytes": (_Z, [C.POINTER(WgradDesc)]),
    "fcd_conv_wgrad": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _Z, _P]),
    "fcd_conv_wgrad2_supported": (_I, [C.POINTER(WgradDesc)]),
    "fcd_conv_wgrad2": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "fcd_pack_weight_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fcd_pack_weight_adj": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_conv_wants_split_weights": (_I, [C.POINTER(ConvDesc)]),
    "fcd_conv_weight_format": (_I, [C.POINTER(ConvDesc)]),
    "fcd_pack_weight_fwd_h2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fcd_pack_weight_adj_h2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_pack_weight_fwd_h2_batch": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_pack_weight_adj_h2_batch": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_pack_weight_fwd_x3": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fcd_pack_weight_adj_x3": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_head_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_head_dgrad": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_head_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_instnorm_workspace_bytes": (_Z, [_I, _I, _I]),
    "fcd_instnorm_stats": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _Z, _P]),
    "fcd_instnorm_stats_noalloc": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "fcd_instnorm_stats_from_running": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "fcd_instnorm_update_running": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "fcd_instnorm_update_running_batch": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "fcd_instnorm_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "fcd_instnorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "fcd_absmax": (_I, [_P, _Z, _P, _P]),
    "fcd_absmax_batch": (_I, [_P, _P, _I, _P, _P]),
    "fcd_instnorm_apply_amax": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P]),
    "fcd_instnorm_bwd_amax": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _Z, _P]),
    "fcd_instnorm_bwd_s12": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _Z, _P]),
    "fcd_instnorm_affine_batch": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "fcd_g_prologue": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_g_epilogue_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fcd_g_epilogue_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fcd_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcd_lrelu_bwd": (_I, [_P, _P, _P, _Z, _F, _P]),
    "fcd_act_fwd": (_I, [_P, _P, _Z, _F, _P]),
    "fcd_act_bwd": (_I, [_P, _P, _P, _Z, _F, _I, _P]),
    "fcd_act_bwd2": (_I, [_P, _P, _P, _P, _P, _Z, _F, _P]),
    "fcd_colsum_workspace_bytes": (_Z, [_Z, _I]),
    "fcd_colsum": (_I, [_P, _P, _I, _Z, _I, _P, _Z, _P]),
    "fcd_reduce_workspace_bytes": (_Z, []),
    "fcd_sum_scaled": (_I, [_P, _P, _Z, _F, _P, _Z, _P]),
    "fcd_fill_scaled": (_I, [_P, _Z, _F, _P, _P]),
    "fcd_l1_mean": (_I, [_P, _P, _P, _Z, _P, _Z, _P]),
    "fcd_l1_mean_bwd": (_I, [_P, _P, _P, _P, _P, _I, _Z, _P]),
    "fcd_bce_logits_mean": (_I, [_P, _P, _P, _Z, _P, _Z, _P]),
    "fcd_bce_logits_mean_bwd": (_I, [_P, _P, _P, _P, _Z, _P]),
    "fcd_lerp_rows": (_I, [_P, _P, _P, _P, _I, _Z, _P]),
    "fcd_gp_fwd": (_I, [_P, _P, _P, _I, _Z, _P]),
    "fcd_gp_bwd": (_I, [_P, _P, _P, _P, _I, _Z, _P]),
    "fcd_adam_step": (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _I, _P]),
    "fcd_diffmap_threshold": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _I, _P]),
    "fcd_scene_tiles": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "fcd_scene_tile_valid": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "fcd_scene_tif_bytes": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "fcd_pseudo_mask": (_I, [_P, _P, C.c_size_t, _F, _P, _P, _P]),
    "fcd_scene_stitch": (_I, [_P, _P, _I, _I, _P, _I, _I, _P]),
    "fcd_scene_binarize": (_I, [_P, _P, _I, _I, _I, C.c_double, _P, _P]),
    "fcd_patch_batch": (_I, [_P, C.c_long, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "fcd_valid_minmax": (_I, [_P, _P, C.c_size_t, _P, _P]),
    "fcd_threshold_hist": (_I, [_P, _P, C.c_size_t, _P, _I, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library once; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"fcd_amd: {LIB_PATH} is missing. Build it with `make` (hipcc --offload-arch=gfx950); "
            "there is no CPU or stock-PyTorch fallback for the FCD hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.fcd_abi_version() != ABI_VERSION:
        raise RuntimeError(f"fcd_amd: libfcd_hip.so has ABI version {lib.fcd_abi_version()}, this binding needs {ABI_VERSION}")
    if lib.fcd_conv_desc_bytes() != C.sizeof(ConvDesc) or lib.fcd_wgrad_desc_bytes() != C.sizeof(WgradDesc):
        raise RuntimeError("fcd_amd: descriptor layouts of libfcd_hip.so and of this binding differ "
                           f"(FcdConvDesc {lib.fcd_conv_desc_bytes()} / {C.sizeof(ConvDesc)} bytes, "
                           f"FcdWgradDesc {lib.fcd_wgrad_desc_bytes()} / {C.sizeof(WgradDesc)} bytes)")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().fcd_last_error().decode(errors="replace")
        raise RuntimeError(f"libfcd_hip {what} failed (status {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]):
    """Device address of a contiguous fp32 tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    """The current HIP stream of the current device as a raw handle.  (torch.cuda.current_stream() builds a Stream
    object through several layers of Python per call -- 8 us, some 300 times a step; this is the call under it.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _Workspace:
    """One growing byte buffer per device and stream; calls on one stream are ordered, so it is reused."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        # one buffer per (device, stream): work on the side stream (ops.SideWork) must not share slabs with the chain
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        if idx is None:
            idx = torch._C._cuda_getDevice()
        key = (idx, torch._C._cuda_getCurrentRawStream(idx))
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


workspace = _Workspace()


def require_gpu_tensor(t: torch.Tensor, name: str, dtype: torch.dtype = torch.float32) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"fcd_amd: {name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise RuntimeError(f"fcd_amd: {name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"fcd_amd: {name} must be contiguous")
