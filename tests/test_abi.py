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
