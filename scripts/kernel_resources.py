#!/usr/bin/env python3
"""Per-kernel resource table of the BUILT kernel library (no GPU needed): VGPRs, AGPRs, SGPRs, scratch (private segment) and
static LDS of every gfx950 kernel in whisper.cpp_amd/lib/libmi355x_kernels.so, read from the code objects' metadata.
  python scripts/kernel_resources.py [--json]        (tests/test_host.py asserts on it; profiles/archive/r01_kernel_resources.txt)
"""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(so: Path):
    out = []
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(so), str(d / "fat.bin")], check=True)
        data = (d / "fat.bin").read_bytes()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):      # one bundle per translation unit
            (d / f"b{n}.bin").write_bytes(data[a:b])
            subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={d / f'b{n}.bin'}", f"--output={d / f'b{n}.co'}"], check=True, stderr=subprocess.DEVNULL)
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(d / f"b{n}.co")], stdout=subprocess.PIPE, text=True).stdout
            cur = None
            in_kernels = False
            for line in notes.splitlines():
                if re.match(r"\s*amdhsa\.kernels:", line):
                    in_kernels = True
                    continue
                if not in_kernels:
                    continue
                if re.match(r"\s*amdhsa\.\w+:", line):              # next top-level key (amdhsa.target, amdhsa.version)
                    in_kernels = False
                    continue
                if re.match(r"^  - \.", line):                       # a new entry of the kernel list (keys are sorted: .agpr_count first)
                    if cur and "name" in cur:
                        out.append(cur)
                    cur = {}
                m = re.match(r"^  [- ] \.(name|vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|max_flat_workgroup_size):\s+(\S+)", line)
                if m and cur is not None:
                    cur[m.group(1)] = m.group(2) if m.group(1) == "name" else int(m.group(2))
            if cur and "name" in cur:
                out.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out) + "\n", stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["demangled"] = n.replace("void ", "").split("(")[0]
    return out


def disassemble(so: Path, mangled: str) -> str:
    """ISA of one kernel of the built library (llvm-objdump on the code object that defines it)"""
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(so), str(d / "fat.bin")], check=True)
        data = (d / "fat.bin").read_bytes()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
            (d / f"b{n}.bin").write_bytes(data[a:b])
            subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={d / f'b{n}.bin'}", f"--output={d / f'b{n}.co'}"], check=True, stderr=subprocess.DEVNULL)
            syms = subprocess.run([str(LLVM / "llvm-readelf"), "-s", "-W", str(d / f"b{n}.co")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
            if re.search(rf"\s{re.escape(mangled)}$", syms, re.M):
                return subprocess.run([str(LLVM / "llvm-objdump"), "-d", f"--disassemble-symbols={mangled}", str(d / f"b{n}.co")],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    return ""


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--disasm":
        print(disassemble(ROOT / "whisper.cpp_amd" / "lib" / "libmi355x_kernels.so", sys.argv[2]))
        sys.exit(0)
    ks = kernels(ROOT / "whisper.cpp_amd" / "lib" / "libmi355x_kernels.so")
    if "--json" in sys.argv:
        print(json.dumps(ks))
    else:
        print(f"{len(ks)} gfx950 kernels; with scratch: {sum(1 for k in ks if k.get('private_segment_fixed_size', 0) > 0)}")
        print(f"{'kernel':64s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>7s} {'wg':>5s}")
        for k in sorted(ks, key=lambda k: k["demangled"]):
            print(f"{k['demangled'][:64]:64s} {k.get('vgpr_count', 0):5d} {k.get('agpr_count', 0):5d} {k.get('sgpr_count', 0):5d} "
                  f"{k.get('private_segment_fixed_size', 0):8d} {k.get('group_segment_fixed_size', 0):7d} {k.get('max_flat_workgroup_size', 0):5d}")
