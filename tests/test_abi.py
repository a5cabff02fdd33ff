"""The drop-in boundary: both shared libraries load (no GPU needed) and export every symbol include/*.h declares."""
import ctypes as C
import re
from pathlib import Path

import pytest

from conftest import ROOT, has_gpu

KERN = ROOT / "whisper.cpp_amd" / "lib" / "libmi355x_kernels.so"
PLUG = ROOT / "whisper.cpp_amd" / "lib" / "libggml-mi355x.so"
REFBASE = ROOT / "oracle" / "_ref" / "libggml-base.so"


def declared(header: str, macro: str):
    txt = (ROOT / "include" / header).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith("#"))
    return sorted(set(re.findall(macro + r"\s+[^;(]*?\b(\w+)\s*\(", txt)))


def test_kernel_library_exports_its_header():
    names = declared("mi355x_kernels.h", "MI355X_API")
    assert len(names) >= 30
    lib = C.CDLL(str(KERN))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    from whisper_cpp_amd import kernels_api
    assert set(kernels_api.SYMBOLS) <= set(names)


def test_activation_layout_sizes_are_host_functions_of_the_header():
    """include/mi355x_kernels.h: sizes of the two quantized-activation layouts, computed on the host (no device needed).
    PLANES (decode): 40 bytes per 32 features and column for the Q8_0 family, K + 4 K/256 + 4 K/32 per column for Q8_K (Q4_K weights); more than 8
    columns = ceil(T/8) images of 8 columns back to back, image g at g * bytes(8 columns).  ROWS (wide products, csrc/kernels/qrows.h): int8 q[T][K] plus
    one f32 scale per 32 (Q8_0) or per 256 elements and one i32 sum per 32 (Q8_K)."""
    lib = C.CDLL(str(KERN))
    lib.mi355x_act_planes_bytes.restype = C.c_size_t
    lib.mi355x_act_planes_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.mi355x_act_rows_bytes.restype = C.c_size_t
    lib.mi355x_act_rows_bytes.argtypes = [C.c_int, C.c_int64, C.c_int64]
    Q5_0, Q4_K = 6, 12

    def al16(n):
        return (n + 15) // 16 * 16
    for K in (384, 1280, 5120):
        full = lib.mi355x_act_planes_bytes(Q5_0, K, 8)
        assert full == 8 * (K // 32) * 40
        for T in (1, 3, 8):
            assert lib.mi355x_act_planes_bytes(Q5_0, K, T) == al16(T * (K // 32) * 40)
        for T in (9, 12, 16, 17, 29, 32):
            g = (T + 7) // 8
            assert lib.mi355x_act_planes_bytes(Q5_0, K, T) == (g - 1) * full + al16((T - 8 * (g - 1)) * (K // 32) * 40), (K, T)
        assert lib.mi355x_act_rows_bytes(Q5_0, K, 1500) == 1500 * K + 1500 * (K // 32) * 4
    K = 1280
    per_col = K + (K // 256) * 4 + (K // 32) * 4
    assert lib.mi355x_act_planes_bytes(Q4_K, K, 8) == 8 * per_col
    assert lib.mi355x_act_planes_bytes(Q4_K, K, 11) == 8 * per_col + al16(3 * per_col)
    assert lib.mi355x_act_rows_bytes(Q4_K, K, 1500) == 1500 * K + 1500 * (K // 256) * 4 + 1500 * (K // 32) * 4


def test_plugin_exports_ggml_entry_points():
    if not REFBASE.exists():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    C.CDLL(str(REFBASE), mode=C.RTLD_GLOBAL)
    C.CDLL(str(KERN), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(str(PLUG))
    names = declared("ggml_mi355x.h", "GGML_MI355X_API")
    assert "ggml_backend_init" in names and "ggml_backend_score" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.ggml_backend_init.restype = C.c_void_p
    reg = lib.ggml_backend_init()
    assert reg, "ggml_backend_init returned NULL"
    # struct ggml_backend_reg { int api_version; ... } — must be GGML_BACKEND_API_VERSION (2) or the loader rejects it
    assert C.cast(reg, C.POINTER(C.c_int))[0] == 2
    score = lib.ggml_backend_score()
    assert (score > 0) == has_gpu()


def test_reference_loader_accepts_or_skips_plugin(plugin_env):
    """The unmodified reference binary must not crash with GGML_BACKEND_PATH set, GPU or not."""
    import subprocess
    exe = ROOT / "oracle" / "_ref" / "whisper-bench"
    if not exe.exists():
        pytest.skip("oracle/_ref not built")
    r = subprocess.run([str(exe), "-m", "/nonexistent.bin"], env=plugin_env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert "load_backend" in r.stdout                       # the loader saw the plugin
    assert ("not supported on this system" in r.stdout) != has_gpu()


def test_ctypes_mirrors_have_the_layout_of_the_c_structs(tmp_path):
    """the ctypes structures of whisper.cpp_amd/kernels_api.py against sizeof / offsetof of include/mi355x_kernels.h as gcc lays them
    out, and positional construction of the epilogue in its documented order (round 3: `bias_per_col` took the padding slot behind
    `gelu`; a mirror that had shifted `residual` by one position crashed the GPU suite)"""
    import subprocess
    from whisper_cpp_amd import kernels_api as ka
    probes = {
        "mi355x_epilogue": ["bias", "scale", "has_scale", "gelu", "bias_per_col", "residual", "residual_nb1"],
        "mi355x_tensor": ["data", "type", "ne", "nb"],
        "mi355x_gemv_seg": ["w", "wtype", "N", "ep", "dst", "dst_type", "dst_nb1"],
        "mi355x_gemv_desc": ["x", "x_nb1", "K", "T", "has_norm", "eps", "ln_w", "ln_b", "nseg", "seg", "attn_part_o", "attn_part_ml", "attn_nparts", "x_planes", "planes_out", "planes_out_only", "cols"],
        "mi355x_attn_partials": ["part_o", "part_ml", "nparts", "T", "H"],
    }
    mirrors = {"mi355x_epilogue": ka.Epilogue, "mi355x_tensor": ka.Tensor, "mi355x_gemv_seg": ka.GemvSeg, "mi355x_gemv_desc": ka.GemvDesc, "mi355x_attn_partials": ka.AttnPartials}
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "mi355x_kernels.h"', "int main(void) {"]
    for s, fields in probes.items():
        src.append(f'  printf("{s} %zu", sizeof({s}));')
        for f in fields:
            src.append(f'  printf(" %zu", offsetof({s}, {f}));')
        src.append('  printf("\\n");')
    src += ["  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", f"-I{ROOT / 'include'}", str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout
    for line in out.strip().splitlines():
        name, size, *offs = line.split()
        m = mirrors[name]
        assert C.sizeof(m) == int(size), (name, C.sizeof(m), size)
        for f, o in zip(probes[name], offs):
            assert getattr(m, f).offset == int(o), (name, f, getattr(m, f).offset, o)
    e = ka.Epilogue(11, 0.5, 1, 1, 22, 33)                                  # bias, scale, has_scale, gelu, residual, residual_nb1
    assert (e.bias, e.scale, e.has_scale, e.gelu, e.bias_per_col, e.residual, e.residual_nb1) == (11, 0.5, 1, 1, 0, 22, 33)
    e = ka.Epilogue(bias=5, bias_per_col=1)
    assert e.bias == 5 and e.bias_per_col == 1 and e.residual is None


@pytest.mark.skipif(has_gpu(), reason="the no-device failure path")
def test_in_process_rccl_broadcast_fails_cleanly_without_a_device():
    """ggml_backend_mi355x_broadcast_weights_rccl_group on a machine without a gfx950 device: librccl is found and bound (or reported missing), the call returns a
    negative code with stats zeroed, nothing crashes and nothing is written to the host's stdout — the path bench.py's peer fall-back and its rccl_error field rest on."""
    import subprocess, sys
    code = (
        "import ctypes as C, sys\n"
        f"C.CDLL(r'{REFBASE}', mode=C.RTLD_GLOBAL)\n"
        f"p = C.CDLL(r'{PLUG}')\n"
        "p.ggml_backend_mi355x_broadcast_weights_rccl_group.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double)]\n"
        "dev = (C.c_int * 2)(0, 1); st = (C.c_double * 6)(*([7.0] * 6))\n"
        "rc = p.ggml_backend_mi355x_broadcast_weights_rccl_group(dev, 2, st)\n"
        "dup = p.ggml_backend_mi355x_broadcast_weights_rccl_group((C.c_int * 2)(0, 0), 2, st)\n"
        "sys.stderr.write('rc=%d dup=%d stats=%s\\n' % (rc, dup, list(st)))\n"
        "sys.exit(0 if rc < 0 and dup < 0 and not any(st) else 1)\n")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert r.stdout == "", r.stdout[:400]
