"""ctypes binding of include/mi355x_kernels.h (the kernel library's C ABI).

Used by tests/ and bench.py to drive the HIP kernels directly with device memory owned by PyTorch-ROCm
(torch is only the allocator / stream plumbing here).  Fails loudly when the library is missing.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import KERNELS_SO

F32, F16, Q4_0, Q5_0, Q8_0, Q4_K, I32 = 0, 1, 2, 6, 8, 12, 26
TYPE_NAMES = {"f32": F32, "f16": F16, "q4_0": Q4_0, "q5_0": Q5_0, "q8_0": Q8_0, "q4_k": Q4_K, "i32": I32}
_BLOCK = {F32: (1, 4), F16: (1, 2), I32: (1, 4), Q4_0: (32, 18), Q5_0: (32, 22), Q8_0: (32, 34), Q4_K: (256, 144)}

# every symbol declared in include/mi355x_kernels.h (tests/test_abi.py checks header <-> library <-> this list)
SYMBOLS = [
    "mi355x_device_count", "mi355x_ctx_create", "mi355x_ctx_destroy", "mi355x_ctx_stream", "mi355x_ctx_synchronize", "mi355x_flush", "mi355x_norm_prep", "mi355x_flash_attn_ext_prep", "mi355x_gemm_f16act_prep",
    "mi355x_last_error", "mi355x_eager_count", "mi355x_prof_enable", "mi355x_prof_report",
    "mi355x_prof_reset", "mi355x_type_is_quantized", "mi355x_type_row_bytes", "mi355x_repack_to_planar",
    "mi355x_repack_from_planar", "mi355x_mul_mat", "mi355x_prep_act", "mi355x_gemm_f16act", "mi355x_dequant_f16", "mi355x_gemv_fused",
    "mi355x_flash_attn_ext", "mi355x_flash_attn_ext_exact", "mi355x_flash_attn_partial", "mi355x_flash_attn_combine", "mi355x_norm", "mi355x_binary", "mi355x_scale", "mi355x_gelu", "mi355x_cpy",
    "mi355x_last_launch_mirrored", "mi355x_act_planes_bytes", "mi355x_act_prepare", "mi355x_act_scratch", "mi355x_flash_attn_partial_multi", "mi355x_flash_attn_planes", "mi355x_decode_head_multi",
    "mi355x_argmax_top2", "mi355x_act_rows_bytes", "mi355x_gemm_q8act", "mi355x_gemm_q8act_prep", "mi355x_flash_attn_ext_prep_rows", "mi355x_unary", "mi355x_pad_reflect_1d", "mi355x_get_rows", "mi355x_get_rows_add", "mi355x_im2col_1d", "mi355x_soft_max", "mi355x_rope", "mi355x_concat", "mi355x_memset", "mi355x_checksum", "mi355x_log_mel", "mi355x_log_mel_n_len", "mi355x_debug_read_stamps", "mi355x_wake", "mi355x_test_option",
]


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int32), ("reserved", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4)]


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("scale", C.c_float), ("has_scale", C.c_int32), ("gelu", C.c_int32), ("bias_per_col", C.c_int32),
                ("residual", C.c_void_p), ("residual_nb1", C.c_int64)]

    def __init__(self, bias=None, scale=0.0, has_scale=0, gelu=0, residual=None, residual_nb1=0, bias_per_col=0):
        # positional order of the struct BEFORE bias_per_col took the padding slot behind `gelu` (round 3): callers that fill the
        # struct positionally keep meaning what they meant
        super().__init__()
        self.bias, self.scale, self.has_scale, self.gelu = bias or None, scale, has_scale, gelu
        self.residual, self.residual_nb1, self.bias_per_col = residual or None, residual_nb1, bias_per_col


class GemvSeg(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wtype", C.c_int32), ("N", C.c_int32), ("ep", Epilogue), ("dst", C.c_void_p),
                ("dst_type", C.c_int32), ("reserved", C.c_int32), ("dst_nb1", C.c_int64)]


class GemvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_nb1", C.c_int64), ("K", C.c_int32), ("T", C.c_int32), ("has_norm", C.c_int32),
                ("eps", C.c_float), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("nseg", C.c_int32), ("reserved", C.c_int32),
                ("seg", GemvSeg * 3), ("attn_part_o", C.c_void_p), ("attn_part_ml", C.c_void_p), ("attn_nparts", C.c_int32),
                ("reserved2", C.c_int32), ("x_planes", C.c_void_p), ("planes_out", C.c_void_p), ("planes_out_only", C.c_int32),
                ("reserved3", C.c_int32), ("cols", C.c_void_p)]


MAX_COLS = 32          # mi355x_kernels.h: MI355X_MAX_COLS (more than IMG_COLS = 8 columns travel as images of 8)
IMG_COLS = 8


class GemvCols(C.Structure):          # mi355x_gemv_cols: [segment][column] destinations / residuals
    _fields_ = [("dst", (C.c_void_p * MAX_COLS) * 3), ("res", (C.c_void_p * MAX_COLS) * 3), ("mirror", C.c_void_p * MAX_COLS)]


class ActDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_nb1", C.c_int64), ("xcol", C.c_void_p * MAX_COLS), ("K", C.c_int32), ("T", C.c_int32),
                ("wtype", C.c_int32), ("has_norm", C.c_int32), ("eps", C.c_float), ("reserved", C.c_int32), ("ln_w", C.c_void_p),
                ("ln_b", C.c_void_p), ("attn_part_o", C.c_void_p), ("attn_part_ml", C.c_void_p), ("attn_nparts", C.c_int32),
                ("reserved2", C.c_int32)]


class AttnState(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("mask", C.c_void_p), ("n_kv", C.c_int32), ("reserved", C.c_int32)]


class HeadState(C.Structure):
    _fields_ = [("tok", C.c_void_p), ("pos", C.c_void_p), ("dst", C.c_void_p), ("mask_f32", C.c_void_p), ("mask_f16", C.c_void_p),
                ("n_mask", C.c_int32), ("reserved", C.c_int32)]


class AttnPartials(C.Structure):
    _fields_ = [("part_o", C.c_void_p), ("part_ml", C.c_void_p), ("nparts", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("reserved", C.c_int32)]


class RopeParams(C.Structure):
    _fields_ = [("n_dims", C.c_int32), ("mode", C.c_int32), ("n_ctx_orig", C.c_int32), ("freq_base", C.c_float),
                ("freq_scale", C.c_float), ("ext_factor", C.c_float), ("attn_factor", C.c_float), ("beta_fast", C.c_float),
                ("beta_slow", C.c_float), ("sections", C.c_int32 * 4)]


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char_p), ("calls", C.c_uint64), ("total_ms", C.c_double), ("algo_bytes", C.c_double), ("algo_flops", C.c_double)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not Path(KERNELS_SO).exists():
            raise RuntimeError(f"{KERNELS_SO} is missing: run `python whisper.cpp_amd/build.py` (no CPU fallback exists)")
        L = C.CDLL(str(KERNELS_SO), mode=C.RTLD_GLOBAL)
        L.mi355x_ctx_create.restype = C.c_void_p
        L.mi355x_ctx_create.argtypes = [C.c_int]
        L.mi355x_ctx_stream.restype = C.c_void_p
        L.mi355x_last_error.restype = C.c_char_p
        L.mi355x_type_row_bytes.restype = C.c_size_t
        L.mi355x_type_row_bytes.argtypes = [C.c_int, C.c_int64]
        for name in ("mi355x_ctx_destroy", "mi355x_ctx_synchronize", "mi355x_ctx_stream", "mi355x_prof_reset", "mi355x_flush", "mi355x_eager_count", "mi355x_last_launch_mirrored"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.mi355x_eager_count.restype = C.c_uint64
        L.mi355x_prof_enable.argtypes = [C.c_void_p, C.c_int]
        L.mi355x_prof_report.argtypes = [C.c_void_p, C.POINTER(ProfRow), C.c_int]
        L.mi355x_repack_to_planar.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.mi355x_repack_from_planar.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        TP = C.POINTER(Tensor)
        L.mi355x_mul_mat.argtypes = [C.c_void_p, TP, TP, TP, C.POINTER(Epilogue)]
        L.mi355x_prep_act.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int]
        L.mi355x_gemm_f16act.argtypes = [C.c_void_p, TP, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.POINTER(Epilogue)]
        L.mi355x_gemm_f16act_prep.argtypes = [C.c_void_p, TP, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(Epilogue), C.c_void_p]
        L.mi355x_dequant_f16.argtypes = [C.c_void_p, TP, C.c_void_p]
        L.mi355x_act_rows_bytes.restype = C.c_size_t
        L.mi355x_act_rows_bytes.argtypes = [C.c_int, C.c_int64, C.c_int64]
        L.mi355x_gemm_q8act.argtypes = [C.c_void_p, TP, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.POINTER(Epilogue)]
        L.mi355x_gemm_q8act_prep.argtypes = [C.c_void_p, TP, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(Epilogue), C.c_void_p]
        L.mi355x_flash_attn_ext_prep_rows.argtypes = [C.c_void_p, TP, TP, TP, TP, TP, C.c_float, C.c_void_p]
        L.mi355x_gemv_fused.argtypes = [C.c_void_p, C.POINTER(GemvDesc)]
        L.mi355x_flash_attn_ext.argtypes = [C.c_void_p, TP, TP, TP, TP, TP, C.c_float]
        L.mi355x_flash_attn_ext_prep.argtypes = [C.c_void_p, TP, TP, TP, TP, TP, C.c_float, C.c_void_p]
        L.mi355x_flash_attn_ext_exact.argtypes = [C.c_void_p, TP, TP, TP, TP, TP, C.c_float, C.c_int]
        L.mi355x_flash_attn_partial.argtypes = [C.c_void_p, TP, TP, TP, TP, C.c_float, C.POINTER(AttnPartials)]
        L.mi355x_flash_attn_combine.argtypes = [C.c_void_p, C.POINTER(AttnPartials), TP]
        L.mi355x_act_planes_bytes.restype = C.c_size_t
        L.mi355x_act_planes_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.mi355x_act_prepare.argtypes = [C.c_void_p, C.POINTER(ActDesc), C.c_void_p]
        L.mi355x_act_scratch.restype = C.c_void_p
        L.mi355x_act_scratch.argtypes = [C.c_void_p, C.c_int]
        L.mi355x_flash_attn_partial_multi.argtypes = [C.c_void_p, C.c_int, C.POINTER(AttnState), TP, TP, TP, C.c_float, C.POINTER(AttnPartials)]
        L.mi355x_flash_attn_planes.argtypes = [C.c_void_p, C.c_int, C.POINTER(AttnState), TP, TP, TP, C.c_float, C.c_void_p]
        L.mi355x_decode_head_multi.argtypes = [C.c_void_p, C.c_int, C.POINTER(HeadState), TP, TP]
        L.mi355x_norm.argtypes = [C.c_void_p, TP, TP, C.c_float, C.c_void_p, C.c_void_p]
        L.mi355x_norm_prep.argtypes = [C.c_void_p, TP, TP, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mi355x_binary.argtypes = [C.c_void_p, C.c_int, TP, TP, TP]
        L.mi355x_scale.argtypes = [C.c_void_p, TP, TP, C.c_float, C.c_float]
        L.mi355x_gelu.argtypes = [C.c_void_p, TP, TP]
        L.mi355x_argmax_top2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mi355x_unary.argtypes = [C.c_void_p, C.c_int, TP, TP]
        L.mi355x_pad_reflect_1d.argtypes = [C.c_void_p, TP, TP, C.c_int, C.c_int]
        L.mi355x_cpy.argtypes = [C.c_void_p, TP, TP]
        L.mi355x_get_rows.argtypes = [C.c_void_p, TP, TP, TP]
        L.mi355x_get_rows_add.argtypes = [C.c_void_p, TP, TP, TP, TP, TP]
        L.mi355x_im2col_1d.argtypes = [C.c_void_p, TP, TP, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mi355x_soft_max.argtypes = [C.c_void_p, TP, TP, TP, C.c_float, C.c_float]
        L.mi355x_rope.argtypes = [C.c_void_p, TP, TP, C.c_void_p, TP, C.POINTER(RopeParams)]
        L.mi355x_concat.argtypes = [C.c_void_p, TP, TP, TP, C.c_int]
        L.mi355x_checksum.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.mi355x_log_mel_n_len.argtypes = [C.c_int]
        L.mi355x_log_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.mi355x_memset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]
        L.mi355x_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def row_bytes(t: int, ne0: int) -> int:
    blk, sz = _BLOCK[t]
    assert ne0 % blk == 0
    return ne0 // blk * sz


def tensor(ptr: int, t: int, ne, nb=None) -> Tensor:
    """Describe a device tensor.  ne: up to 4 dims (ggml order, fastest first); nb: byte strides (default contiguous)."""
    ne = list(ne) + [1] * (4 - len(ne))
    if nb is None:
        nb = [_BLOCK[t][1], row_bytes(t, ne[0])]
        nb.append(nb[1] * ne[1])
        nb.append(nb[2] * ne[2])
    nb = list(nb) + [0] * (4 - len(nb))
    m = Tensor()
    m.data, m.type = ptr, t
    for i in range(4):
        m.ne[i], m.nb[i] = ne[i], nb[i]
    return m


def repack_to_planar(t: int, blocks: np.ndarray, nelements: int) -> np.ndarray:
    src = np.ascontiguousarray(blocks).view(np.uint8).ravel()
    dst = np.empty_like(src)
    rc = lib().mi355x_repack_to_planar(t, src.ctypes.data, dst.ctypes.data, nelements)
    if rc != 0:
        raise RuntimeError(f"repack_to_planar failed: {rc}")
    return dst


def repack_from_planar(t: int, planar: np.ndarray, nelements: int) -> np.ndarray:
    src = np.ascontiguousarray(planar).view(np.uint8).ravel()
    dst = np.empty_like(src)
    rc = lib().mi355x_repack_from_planar(t, src.ctypes.data, dst.ctypes.data, nelements)
    if rc != 0:
        raise RuntimeError(f"repack_from_planar failed: {rc}")
    return dst


class Ctx:
    """One kernel context (HIP stream + scratch).  Raises if no gfx950 device is usable."""

    def __init__(self, device: int = 0):
        self.h = lib().mi355x_ctx_create(device)
        if not self.h:
            raise RuntimeError("mi355x_ctx_create failed: " + (lib().mi355x_last_error() or b"").decode())

    def sync(self):
        rc = lib().mi355x_ctx_synchronize(self.h)
        if rc:
            raise RuntimeError(f"sync failed rc={rc}: {lib().mi355x_last_error().decode()}")

    def check(self, rc: int, what: str = "op"):
        if rc != 0:
            raise RuntimeError(f"{what} failed rc={rc}: {(lib().mi355x_last_error() or b'').decode()}")

    def close(self):
        if self.h:
            lib().mi355x_ctx_destroy(self.h)
            self.h = None

    def prof(self, on: bool):
        lib().mi355x_prof_enable(self.h, 1 if on else 0)

    def prof_reset(self):
        lib().mi355x_prof_reset(self.h)

    def prof_report(self):
        rows = (ProfRow * 64)()
        n = lib().mi355x_prof_report(self.h, rows, 64)
        return [dict(name=rows[i].name.decode(), calls=rows[i].calls, total_ms=rows[i].total_ms,
                     algo_bytes=rows[i].algo_bytes, algo_flops=rows[i].algo_flops) for i in range(n)]
