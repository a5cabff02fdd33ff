import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))          # tooling used by the tests (synthetic model files)

import __graft_entry__ as graft  # noqa: E402

graft.load_package()

# the suite is self-sufficient on a fresh checkout: the native libraries are git-ignored build products (hipcc cross-compiles
# without a GPU; ~3 min once).  On the GPU box the prebuilt files travel with the snapshot, so this never triggers there.
if not (ROOT / "whisper.cpp_amd" / "lib" / "libmi355x_kernels.so").exists() or not (ROOT / "whisper.cpp_amd" / "lib" / "libggml-mi355x.so").exists():
    graft.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/liboracle.so (the CPU restatement; test infrastructure)."""
    so = ROOT / "oracle" / "liboracle.so"
    if not so.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "oracle"], check=True, stdout=subprocess.DEVNULL)
    L = C.CDLL(str(so))
    vp, i64, f32, i32 = C.c_void_p, C.c_int64, C.c_float, C.c_int
    L.oracle_f32_to_f16.restype = C.c_uint16
    L.oracle_f32_to_f16.argtypes = [f32]
    L.oracle_f16_to_f32.restype = f32
    L.oracle_f16_to_f32.argtypes = [C.c_uint16]
    L.oracle_row_size.restype = C.c_size_t
    L.oracle_row_size.argtypes = [i32, i64]
    L.oracle_dequantize_row.argtypes = [i32, vp, vp, i64]
    L.oracle_quantize_row_ref.argtypes = [i32, vp, vp, i64]
    L.oracle_quantize_row_q8_0.argtypes = [vp, vp, i64]
    L.oracle_quantize_row_q8_K.argtypes = [vp, vp, i64]
    L.oracle_vec_dot.restype = f32
    L.oracle_vec_dot.argtypes = [i32, i64, vp, vp]
    L.oracle_mul_mat.argtypes = [i32, vp, vp, vp, i64, i64, i64]
    L.oracle_norm.argtypes = [vp, vp, i64, i64, f32]
    L.oracle_gelu.argtypes = [vp, vp, i64]
    L.oracle_soft_max.argtypes = [vp, vp, vp, i64, i64, f32]
    L.oracle_im2col_1d_f16.argtypes = [vp, vp, i64, i64, i64, i32, i32, i32, i32]
    L.oracle_rope.argtypes = [vp, vp, vp, i64, i64, i64, i32, i32, i32, f32, f32, f32, f32, f32, f32]
    L.oracle_flash_attn.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, f32]
    L.oracle_flash_attn_ext.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, i32]
    L.oracle_log_mel_n_len.argtypes = [i64]
    L.oracle_log_mel_n_len.restype = i64
    L.oracle_log_mel.argtypes = [vp, i64, vp, i64, vp]
    L.oracle_v_expf.argtypes = [f32]
    L.oracle_v_expf.restype = f32
    return L


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def nmse(ref: np.ndarray, got: np.ndarray) -> float:
    ref = ref.astype(np.float64).ravel()
    got = got.astype(np.float64).ravel()
    den = float((ref * ref).sum())
    return float(((ref - got) ** 2).sum() / den) if den > 0 else float(((ref - got) ** 2).sum())


@pytest.fixture(scope="session")
def plugin_env():
    env = dict(os.environ)
    env["GGML_MI355X_PLUGIN"] = str(ROOT / "whisper.cpp_amd" / "lib" / "libggml-mi355x.so")
    env["GGML_BACKEND_PATH"] = env["GGML_MI355X_PLUGIN"]
    env["LD_LIBRARY_PATH"] = f"{ROOT / 'oracle' / '_ref'}:{ROOT / 'whisper.cpp_amd' / 'lib'}:" + env.get("LD_LIBRARY_PATH", "")
    return env
