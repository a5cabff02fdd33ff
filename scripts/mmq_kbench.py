#!/usr/bin/env python3
"""Micro benchmark of the int8 tile GEMM over the quantized operands (csrc/kernels/mmq.hip) against the f16 LDS-DMA ring GEMM on f16
weight copies (gemm_mfma.hip) at the large-v3 encoder shapes (1500 columns), through the C ABI, timed with the library's own
hipEvent-bracketed profiler.  Weights rotate through a pool larger than the Infinity Cache (as 32 layers do).

  python scripts/mmq_kbench.py [--iters 30] [--qtype q5_0] [--what fc1,fc2,oproj,qkv]
One line per (case, variant): us per step, TFLOP/s (2*M*K*T per product)."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from whisper_cpp_amd import kernels_api as ka  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--qtype", default="q5_0")
    ap.add_argument("--what", default="fc1,fc2,oproj,qkv", help="fc1,fc2,oproj,qkv,fixed")
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--anatomy", action="store_true", help="time the kernel with parts of its K-step switched off (GGML_MI355X_MMQ_DBG)")
    a = ap.parse_args()
    what = set(a.what.split(","))
    tid = ka.TYPE_NAMES[a.qtype]
    import torch
    ctx = ka.Ctx(0)
    L = ka.lib()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    T, n = a.T, 1280

    def timed(label, variant, envs, fn, flops, iters):
        old = {k: os.environ.get(k) for k in envs}
        for k, v in envs.items():
            os.environ[k] = str(v)
        try:
            for i in range(3):
                rc = fn(i)
                if rc:
                    print(f"{label:8s} {variant:34s} rc={rc} {L.mi355x_last_error()}")
                    return
            ctx.sync()
            ctx.prof(True)
            ctx.prof_reset()
            for i in range(iters):
                fn(i)
            ctx.sync()
            rows = ctx.prof_report()
            ctx.prof(False)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        tot = sum(r["total_ms"] for r in rows)
        calls = sum(r["calls"] for r in rows)
        us = tot * 1e3 / iters
        names = ",".join(sorted({r["name"] for r in rows}))
        print(f"{label:8s} {variant:34s} {us:8.2f} us per step  {flops / us / 1e6:7.1f} TFLOP/s  ({calls // iters} launches: {names})", flush=True)

    def qweight(M, K):
        """planar quantized weight with random quants and a constant scale (timing only)"""
        nb = M * K // (256 if tid == ka.Q4_K else 32)
        nbytes = M * ka.row_bytes(tid, K)
        w = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
        tail = nb * (4 if tid == ka.Q4_K else 2)
        w[nbytes - tail:] = torch.from_numpy(np.full(tail // 2, 0.01, dtype=np.float16).view(np.uint8)).to("cuda:0")
        return w

    def rows_of(x, K):
        r = torch.zeros(L.mi355x_act_rows_bytes(tid, K, T), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ctx.check(L.mi355x_prep_act(ctx.h, x.data_ptr(), K * 4, 0, r.data_ptr(), K, T, 4 if tid == ka.Q4_K else 3), "rows")
        ctx.sync()
        return r

    x1 = torch.randn((T, n), device="cuda:0", generator=g)
    x4 = torch.randn((T, 4 * n), device="cuda:0", generator=g)
    r1, r4 = rows_of(x1, n), rows_of(x4, 4 * n)
    a1, a4 = x1.to(torch.float16), x4.to(torch.float16)
    bias = torch.zeros(4 * n, device="cuda:0")
    res = torch.zeros((T, n), device="cuda:0")
    y4 = torch.zeros((T, 4 * n), device="cuda:0")
    ys = [torch.zeros((T, n), device="cuda:0") for _ in range(3)]
    prow = torch.zeros(L.mi355x_act_rows_bytes(ka.Q5_0, 4 * n, T), dtype=torch.uint8, device="cuda:0")
    pf16 = torch.zeros((T, 4 * n), dtype=torch.float16, device="cuda:0")
    torch.cuda.synchronize()

    MMQ = [("mmq 64x128 (default)", {}), ("mmq 128x64", {"GGML_MI355X_MMQ_TILE": 12864}), ("mmq 64x128, VALU scales", {"GGML_MI355X_MMQ_SCALE_MFMA": 0})]
    if a.anatomy:
        # what a K-step is made of: the kernel with parts of it switched off (results are garbage, only the time counts)
        MMQ = [("mmq 64x128 (default)", {}), ("  no MFMA / fold", {"GGML_MI355X_MMQ_DBG": 1}), ("  no global loads", {"GGML_MI355X_MMQ_DBG": 2}),
               ("  no LDS stores (no unpack)", {"GGML_MI355X_MMQ_DBG": 4}), ("  loads + stores only", {"GGML_MI355X_MMQ_DBG": 1}),
               ("  compute only", {"GGML_MI355X_MMQ_DBG": 6}), ("  barriers only", {"GGML_MI355X_MMQ_DBG": 7})]

    def cases(label, M, K, rows, act16, ep, prep):
        count = max(4, int(400e6 // (M * K * 2)))
        wq = [qweight(M, K) for _ in range(count)]
        dst = y4 if M == 4 * n else ys[0]

        def mmq(i):
            tw = ka.tensor(wq[i % count].data_ptr(), tid, [K, M])
            if prep:
                return L.mi355x_gemm_q8act_prep(ctx.h, C.byref(tw), rows.data_ptr(), T, None, 0, C.byref(ep), prow.data_ptr()) or L.mi355x_flush(ctx.h)
            return L.mi355x_gemm_q8act(ctx.h, C.byref(tw), rows.data_ptr(), T, dst.data_ptr(), M * 4, ka.F32, C.byref(ep)) or L.mi355x_flush(ctx.h)
        for name, e in MMQ:
            timed(label, name, e, mmq, 2.0 * T * M * K, a.iters)
        del wq
        wf = [(torch.randn((M, K), device="cuda:0", generator=g) * K ** -0.5).to(torch.float16) for _ in range(count)]

        def ring(i):
            tw = ka.tensor(wf[i % count].data_ptr(), ka.F16, [K, M])
            if prep:
                rc = L.mi355x_gemm_f16act_prep(ctx.h, C.byref(tw), act16.data_ptr(), K, T, None, 0, C.byref(ep), pf16.data_ptr())
            else:
                rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act16.data_ptr(), K, T, dst.data_ptr(), M * 4, ka.F32, C.byref(ep))
            return rc or L.mi355x_flush(ctx.h)
        timed(label, "f16 ring on an f16 weight copy", {}, ring, 2.0 * T * M * K, a.iters)

    if "fixed" in what:
        # what a launch costs before its first and after its last K-step: one-K-step products (K = 128) and K = 256 for the slope, with the
        # epilogues of the encoder's products (plain F32 store; bias + table GELU + the next GEMM's rows; bias + residual)
        xs = torch.randn((T, 256), device="cuda:0", generator=g)
        for K in (128, 256):
            rk = torch.zeros(L.mi355x_act_rows_bytes(tid, K, T), dtype=torch.uint8, device="cuda:0")
            xk = xs[:, :K].contiguous()
            torch.cuda.synchronize()
            ctx.check(L.mi355x_prep_act(ctx.h, xk.data_ptr(), K * 4, 0, rk.data_ptr(), K, T, 4 if tid == ka.Q4_K else 3), "rows")
            ctx.sync()
            ak = xk.to(torch.float16)
            ep0 = ka.Epilogue()
            epg = ka.Epilogue(); epg.bias, epg.gelu = bias.data_ptr(), 1
            epr = ka.Epilogue(); epr.bias, epr.residual, epr.residual_nb1 = bias.data_ptr(), res.data_ptr(), n * 4
            if tid != ka.Q4_K or K % 256 == 0:
                cases(f"M5120K{K}", 4 * n, K, rk, ak, ep0, False)
                cases(f"+gelu,prep", 4 * n, K, rk, ak, epg, True)
                cases(f"M1280K{K}", n, K, rk, ak, ep0, False)
                cases(f"+bias,res", n, K, rk, ak, epr, False)
    if "fc1" in what:
        ep = ka.Epilogue()
        ep.bias, ep.gelu = bias.data_ptr(), 1
        cases("fc1", 4 * n, n, r1, a1, ep, True)
    if "fc2" in what:
        ep = ka.Epilogue()
        ep.bias, ep.residual, ep.residual_nb1 = bias.data_ptr(), res.data_ptr(), n * 4
        cases("fc2", n, 4 * n, r4, a4, ep, False)
    if "oproj" in what:
        ep = ka.Epilogue()
        ep.bias, ep.residual, ep.residual_nb1 = bias.data_ptr(), res.data_ptr(), n * 4
        cases("oproj", n, n, r1, a1, ep, False)
    if "qkv" in what:
        ep = ka.Epilogue()
        ep.bias = bias.data_ptr()
        count = 96
        wq = [qweight(n, n) for _ in range(count)]

        def qkv(i):
            for j in range(3):
                tw = ka.tensor(wq[(3 * i + j) % count].data_ptr(), tid, [n, n])
                rc = L.mi355x_gemm_q8act(ctx.h, C.byref(tw), r1.data_ptr(), T, ys[j].data_ptr(), n * 4, ka.F32, C.byref(ep))
                if rc:
                    return rc
            return 0
        def qkv_flush(i):
            return qkv(i) or L.mi355x_flush(ctx.h)
        timed("qkv", "mmq, grouped launch", {}, qkv_flush, 3 * 2.0 * T * n * n, a.iters)
        timed("qkv", "mmq, 3 launches", {"GGML_MI355X_MMQ_GROUP": 0}, qkv_flush, 3 * 2.0 * T * n * n, a.iters)
        del wq
        wf = [(torch.randn((n, n), device="cuda:0", generator=g) * n ** -0.5).to(torch.float16) for _ in range(count)]

        def qkv16(i):
            for j in range(3):
                tw = ka.tensor(wf[(3 * i + j) % count].data_ptr(), ka.F16, [n, n])
                rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), a1.data_ptr(), n, T, ys[j].data_ptr(), n * 4, ka.F32, C.byref(ep))
                if rc:
                    return rc
            return L.mi355x_flush(ctx.h)
        timed("qkv", "f16 ring, grouped launch", {}, qkv16, 3 * 2.0 * T * n * n, a.iters)
    ctx.close()


if __name__ == "__main__":
    main()
