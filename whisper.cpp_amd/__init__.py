"""MI355X-native hot path for whisper.cpp: ggml backend plugin + HIP kernel library (gfx950 only).

The directory name contains a dot, so import it through `__graft_entry__.load_package()` (module name
`whisper_cpp_amd`).  Python here is plumbing only (build driver, ctypes bindings of the C ABIs, multi-stream /
multi-rank helpers); the product is `lib/libggml-mi355x.so` + `lib/libmi355x_kernels.so`.
"""
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
LIB_DIR = PKG_DIR / "lib"
KERNELS_SO = LIB_DIR / "libmi355x_kernels.so"
PLUGIN_SO = LIB_DIR / "libggml-mi355x.so"
HOST_DIR = PKG_DIR / "host" / "_whisper"      # the unmodified reference application the plugin drops into
