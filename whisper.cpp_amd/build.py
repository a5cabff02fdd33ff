"""Build the native parts of the MI355X whisper.cpp hot path (gfx950 only, in-tree outputs).

  whisper.cpp_amd/lib/libmi355x_kernels.so : HIP kernels + C ABI (include/mi355x_kernels.h)
  whisper.cpp_amd/lib/libggml-mi355x.so    : ggml backend plugin (include/ggml_mi355x.h); needs the reference's
                                             ggml headers at BUILD time (never copied into this repo) and
                                             oracle/_ref/libggml-base.so to link against.
  oracle/liboracle.so, oracle/_ref/*       : test infrastructure (see oracle/Makefile).

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached by source mtime under build/.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "lib"
BUILD = ROOT / "build"
REF = Path(os.environ.get("WHISPER_REF", "/root/reference"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

KERNEL_SRCS = ["ctx.hip", "elementwise.hip", "gemv.hip", "decode.hip", "decode_q.hip", "decode_mx.hip", "gemm_mfma.hip", "fattn.hip", "fattn_exact.hip", "mel.hip", "mul_mat.hip", "mmq.hip"]
BACKEND_SRCS = ["ggml_mi355x.cpp", "mi_buffers.cpp", "mi_planner.cpp", "mi_batching.cpp", "mi_distribution.cpp"]      # one internal header: mi_backend.h

HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
            "-Wall", "-Wno-unused-function", "-Wno-unused-variable", f"-I{ROOT / 'include'}", f"-I{CSRC / 'kernels'}"]
# mmq.hip: hipcc SLP-packs the fix-up's independent fma chains into v_pk_fma_f32, which costs more than the two scalar operations in the
# gaps between MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
EXTRA_FLAGS = {"mmq.hip": ["-fno-slp-vectorize"]}
if os.environ.get("MI355X_KTIME_BUILD") == "1":      # kernel-anatomy stamps (GGML_MI355X_KTIME=1 at run time, scripts/kbench.py)
    HIPFLAGS.append("-DMI355X_KTIME")


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + r.stdout)
        raise RuntimeError(f"build step failed: {cmd[0]} {cmd[-1]}")
    return r.stdout


def _stale(out: Path, deps):
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps if Path(d).exists())


def build_kernels(verbose=False):
    LIB.mkdir(exist_ok=True)
    (BUILD / "kernels").mkdir(parents=True, exist_ok=True)
    hdrs = [CSRC / "kernels" / "common.h", CSRC / "kernels" / "decode_common.h", CSRC / "kernels" / "qrows.h", ROOT / "include" / "mi355x_kernels.h"]
    objs, jobs = [], []
    for s in KERNEL_SRCS:
        src = CSRC / "kernels" / s
        obj = BUILD / "kernels" / (s + ".o")
        objs.append(obj)
        if _stale(obj, [src] + hdrs):
            jobs.append([HIPCC, *HIPFLAGS, *EXTRA_FLAGS.get(s, []), "-c", str(src), "-o", str(obj)])
    with ThreadPoolExecutor(max_workers=8) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    so = LIB / "libmi355x_kernels.so"
    if jobs or _stale(so, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(so), *map(str, objs)])
    return so


def build_oracle():
    _run(["make", "-C", str(ROOT / "oracle"), "oracle"])
    if (REF / "src" / "whisper.cpp").exists():
        _run(["make", "-C", str(ROOT / "oracle"), "-j8", "ref", f"REF={REF}"])
    return ROOT / "oracle" / "liboracle.so"


def build_backend():
    """ggml backend plugin.  Compiled against the reference's public + backend-impl headers where they lie."""
    so = LIB / "libggml-mi355x.so"
    srcs = [CSRC / "backend" / s for s in BACKEND_SRCS]
    if not (REF / "ggml" / "include" / "ggml.h").exists():
        if so.exists():
            return so  # GPU box: use the prebuilt plugin
        raise RuntimeError(f"{REF} not found and no prebuilt {so}")
    refdir = PKG / "host" / "_whisper"        # the unmodified reference host application (installed by oracle/Makefile)
    deps = srcs + [ROOT / "include" / "ggml_mi355x.h", ROOT / "include" / "mi355x_kernels.h", LIB / "libmi355x_kernels.so"]
    deps += list((CSRC / "backend").glob("*.h"))
    if _stale(so, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
              "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
              f"-I{ROOT / 'include'}", f"-I{REF / 'ggml' / 'include'}", f"-I{REF / 'ggml' / 'src'}",
              *map(str, srcs), "-o", str(so),
              f"-L{LIB}", "-lmi355x_kernels", f"-L{refdir}", "-lggml-base", "-L/opt/rocm/lib", "-lamdhip64",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../host/_whisper"])
    return so


def build_host():
    """native host harness above whisper.h (include/mi355x_host.h): compiled against the reference's PUBLIC headers where they lie,
    linked against the unmodified reference application's libwhisper (host/_whisper)."""
    so = LIB / "libmi355x_host.so"
    src = CSRC / "host" / "host_harness.cpp"
    if not (REF / "include" / "whisper.h").exists():
        if so.exists():
            return so
        raise RuntimeError(f"{REF} not found and no prebuilt {so}")
    refdir = PKG / "host" / "_whisper"
    if _stale(so, [src, ROOT / "include" / "mi355x_host.h"]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-pthread",
              f"-I{ROOT / 'include'}", f"-I{REF / 'include'}", f"-I{REF / 'ggml' / 'include'}", str(src), "-o", str(so),
              f"-L{refdir}", "-lwhisper", "-lggml", "-lggml-base", "-Wl,-rpath,$ORIGIN/../host/_whisper"])
    return so


def build_all(verbose=False):
    k = build_kernels(verbose)
    o = build_oracle()
    b = build_backend()
    build_host()
    return k, o, b


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "kernels":
        print(build_kernels(True))
    elif which == "oracle":
        print(build_oracle())
    elif which == "backend":
        print(build_backend())
    elif which == "host":
        print(build_host())
    else:
        print(build_all(True))
