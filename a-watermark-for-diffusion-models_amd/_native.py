"""ctypes binding of libgswm.so (the C ABI declared in include/gswm.h).

There is NO CPU fallback: if the HIP library has not been built, importing this module's `lib()` raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C a-watermark-for-diffusion-models_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSWM_LIB", os.path.join(_HERE, "libgswm.so"))     # GSWM_LIB: an alternative build of the same ABI (kernel A/Bs)

GSW_F32, GSW_F16, GSW_BF16, GSW_F64 = 0, 1, 2, 3
GSW_OK, GSW_ERR_BAD_ARG, GSW_ERR_UNSUPPORTED, GSW_ERR_RAGGED, GSW_ERR_HIP, GSW_WARN_NO_RECORDS = 0, 1, 2, 3, 4, 5
GSW_EMBED_EXACT_F64, GSW_EMBED_FAST_F32 = 0, 1
GSW_MM_GN_ONLY = 1
GSW_FLAG_SATURATED, GSW_FLAG_NAN = 1, 2
GSW_MSG_INLINE_MAX = 256
GSW_IMG_U8_HWC, GSW_IMG_F16_CHW, GSW_IMG_F32_CHW = 0, 1, 2
GSW_PW_BRIGHTNESS, GSW_PW_CONTRAST, GSW_PW_INVERT, GSW_PW_GRAY, GSW_PW_HFLIP, GSW_PW_VFLIP, GSW_PW_NOISE = range(7)

_u8p = C.POINTER(C.c_uint8)


class GswMmExtras(C.Structure):
    """include/gswm.h: everything an engine launch needs besides its operands (records requested, split-K scratch), and what it did"""
    _fields_ = [("colstats_dev", C.c_void_p), ("colstats_capacity", C.c_int64), ("rowstats_dev", C.c_void_p), ("rowstats_capacity", C.c_int64),
                ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_int64), ("max_splits", C.c_int),
                ("colstats_rows_per_block", C.c_int), ("colstats_blocks", C.c_int), ("rowstats_slots", C.c_int), ("splits", C.c_int), ("flags", C.c_int)]


_PROTOTYPES = {
    "gsw_version": (C.c_int, []),
    "gsw_build_flags": (C.c_int, []),
    "gsw_strerror": (C.c_char_p, [C.c_int]),
    "gsw_last_hip_error": (C.c_int, []),
    "gsw_keystream": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gsw_embed": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                            C.c_int, C.c_int, C.c_int64, C.c_uint32, C.c_void_p]),
    "gsw_philox_uniform": (C.c_int, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "gsw_extract": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int64, C.c_void_p]),
    "gsw_bit_matches": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "gsw_ddim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int64, C.c_void_p]),
    "gsw_ddim_step_cfg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                    C.c_int, C.c_int64, C.c_void_p]),
    "gsw_ddim_step_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_char_p,
                                        C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                        C.c_void_p]),
    "gsw_groupnorm_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "gsw_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "gsw_conv_pf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_groupnorm_pf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_add_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                    C.c_int, C.c_void_p]),
    "gsw_groupnorm_pf2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_conv3x3_res_pf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gsw_conv_up2x_pf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "gsw_attention_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "gsw_xattn_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_xattn_fused_pre": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_gn_colstats_pairs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_gn_proj_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_mt19937_seed": (None, [C.c_uint32, C.c_void_p]),
    "gsw_mt19937_uniform": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gsw_lanczos_plan": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "gsw_resize_lanczos": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "gsw_tensor_to_image": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gsw_jpeg_quant_tables": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "gsw_jpeg_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "gsw_jpeg_roundtrip": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gsw_gaussian_blur_params": (C.c_int, [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsw_gaussian_blur": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsw_image_pointwise": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "gsw_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_int, C.c_void_p]),
    "gsw_gemm_qkv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_ln_rowstats_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "gsw_gemm_ln": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_gemm_strided": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_softmax_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "gsw_mm_config": (C.c_int, [C.c_int, C.c_int]),
    "gsw_mm_get_config": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gsw_groupnorm_pf_cs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p]),
    "gsw_groupnorm_pf_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_gather_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p]),
    "gsw_nchw_to_pf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_conv3x3_pf_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsw_gemm_small_config": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "gsw_gemm_small": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.c_int,
                                 C.c_int, C.c_void_p]),
    "gsw_gemm_ex": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gsw_gemm_ln_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p]),
    "gsw_conv_pf_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gsw_conv3x3_res_pf_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gsw_conv_up2x_pf_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}

_lib = None


class GswError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__(f"libgswm: {what} (status {status})")
        self.status = status


def lib() -> C.CDLL:
    """Load libgswm.so (once).  Raises if it is missing -- the product path never degrades to a CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP library has not been built. Run __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the watermark hot path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        flags = l.gsw_build_flags()
        if flags != 0 and os.environ.get("GSWM_LIB") is None:
            # a side build with measurement switches (csrc/gswm_ablate.inc) computes wrong results by design: only an explicit GSWM_LIB may load one
            raise ImportError(f"{LIB_PATH} was compiled with measurement switches (gsw_build_flags() = {flags:#x}): not a production build; rebuild it "
                              "(__graft_entry__.build()) or select the side build explicitly with GSWM_LIB")
        _lib = l
    return _lib


def exported_symbols():
    return sorted(_PROTOTYPES)


def check(status: int):
    """Map a gsw_status to the exception the reference would raise for the same condition."""
    if status == GSW_OK:
        return
    what = lib().gsw_strerror(status).decode()
    if status == GSW_ERR_RAGGED:
        raise IndexError("string index out of range")  # extract.py:98 on the short trailing segment
    if status == GSW_ERR_BAD_ARG:
        raise ValueError(f"libgswm: {what}")
    if status == GSW_ERR_HIP:
        what += f" (hipError_t {lib().gsw_last_hip_error()})"
    raise GswError(status, what)
