#!/usr/bin/env python3
"""Ring-GEMM scaling probe through the C ABI: one plain product (no epilogue, F32 store) per (M, K, T), timed by the library's hipEvent
profiler; weights rotate through a pool.  Used to separate the per-K-step cost from the fixed cost of a launch and to see how the time
moves with the number of tiles per CU.   python scripts/gemm_probe.py "M,K,T" ...   (process-level environment selects the variant)"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from whisper_cpp_amd import kernels_api as ka  # noqa: E402


def main():
    import torch
    ctx = ka.Ctx(0)
    L = ka.lib()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    iters = 30
    for spec in sys.argv[1:]:
        M, K, T = (int(v) for v in spec.split(","))
        nw = max(2, min(24, int(300e6 / (M * K * 2))))
        ws = [(torch.randn((M, K), device="cuda:0", generator=g) * K ** -0.5).half() for _ in range(nw)]
        act = torch.randn((T, K), device="cuda:0", generator=g).half()
        y = torch.zeros((T, M), device="cuda:0")
        torch.cuda.synchronize()

        def fn(i):
            tw = ka.tensor(ws[i % nw].data_ptr(), ka.F16, [K, M])
            rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), K, T, y.data_ptr(), M * 4, ka.F32, None)
            return rc or L.mi355x_flush(ctx.h)
        for i in range(3):
            fn(i)
        ctx.sync()
        ctx.prof(True)
        ctx.prof_reset()
        for i in range(iters):
            fn(i)
        ctx.sync()
        rows = ctx.prof_report()
        ctx.prof(False)
        us = sum(r["total_ms"] for r in rows) * 1e3 / iters
        t128, t64 = ((M + 127) // 128) * ((T + 127) // 128), ((M + 127) // 128) * ((T + 63) // 64)
        print(f"M {M:5d} K {K:5d} T {T:5d}: {us:8.2f} us  {2.0 * M * K * T / us / 1e6:7.1f} TFLOP/s   K-steps {K // 64:3d}  tiles 128x128 {t128:4d} / 128x64 {t64:4d}", flush=True)
        del ws, act, y
    ctx.close()


if __name__ == "__main__":
    main()
