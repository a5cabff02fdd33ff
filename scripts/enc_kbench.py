#!/usr/bin/env python3
"""Encoder-kernel micro benchmark through the C ABI (include/mi355x_kernels.h): the large-v3 encoder shapes (1500 tokens), each
variant of the attention kernel (key groups per workgroup) and of the LDS-DMA ring GEMM (tile height / width / ring depth) timed with
the library's own hipEvent-bracketed profiler.  Weights rotate through a pool larger than the Infinity Cache, as the 32 layers of an
encoder do; activations are re-used (in the graph they come from the kernel before).

  python scripts/enc_kbench.py [--iters 40] [--what attn,fc1,qkv,xkv,fc2,oproj]
Prints one line per (case, variant): us per launch, TFLOP/s."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from whisper_cpp_amd import kernels_api as ka  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--what", default="attn,fc1,qkv,xkv,fc2,oproj")
    a = ap.parse_args()
    what = set(a.what.split(","))
    import torch
    ctx = ka.Ctx(0)
    L = ka.lib()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    T, n, H, D = 1500, 1280, 20, 64

    def rnd(*shape, dtype=torch.float16, scale=1.0):
        return (torch.randn(shape, device="cuda:0", generator=g) * scale).to(dtype)

    def timed(label, variant, envs, fn, flops, iters):
        old = {k: os.environ.get(k) for k in envs}
        for k, v in envs.items():
            os.environ[k] = str(v)
        try:
            for i in range(3):
                rc = fn(i)
                if rc:
                    print(f"{label:10s} {variant:28s} rc={rc} {L.mi355x_last_error()}")
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
        print(f"{label:10s} {variant:28s} {us:8.2f} us per step  {flops / us / 1e6:7.1f} TFLOP/s  ({calls // iters} launches: {names})", flush=True)

    # ---- attention: 1500 queries x 1500 keys x 20 heads, result + the O-projection's prepared activations
    if "attn" in what:
        q = rnd(T, H, D, dtype=torch.float32, scale=0.6)
        k = rnd(T, H, D, scale=0.6)
        v = rnd(T, H, D)
        o = torch.zeros((T, H, D), device="cuda:0")
        p = torch.zeros((T, H * D), dtype=torch.float16, device="cuda:0")
        tq = ka.tensor(q.data_ptr(), ka.F32, [D, T, H], [4, H * D * 4, D * 4, T * H * D * 4])
        tk = ka.tensor(k.data_ptr(), ka.F16, [D, T, H], [2, H * D * 2, D * 2, T * H * D * 2])
        tv = ka.tensor(v.data_ptr(), ka.F16, [D, T, H], [2, H * D * 2, D * 2, T * H * D * 2])
        to = ka.tensor(o.data_ptr(), ka.F32, [D, H, T])
        torch.cuda.synchronize()
        for ng, tr in ((1, 1), (2, 1), (3, 1), (4, 1), (1, 0), (3, 0)):
            timed("attention", f"key groups {ng}" + ("" if tr else ", V transposed into LDS"), {"GGML_MI355X_FATTN_NG": ng, "GGML_MI355X_FATTN_TR": tr},
                  lambda i: L.mi355x_flash_attn_ext_prep(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), None, C.byref(to), 0.125, p.data_ptr()),
                  4.0 * T * T * D * H, a.iters)

    act = rnd(T, n)
    act4 = rnd(T, 4 * n)
    bias = torch.zeros(4 * n, device="cuda:0")
    res = torch.zeros((T, n), device="cuda:0")
    y = torch.zeros((T, 4 * n), device="cuda:0")
    ys = [torch.zeros((T, n), device="cuda:0") for _ in range(8)]
    prep = torch.zeros((T, 4 * n), dtype=torch.float16, device="cuda:0")

    def pool(M, K, count):
        return [rnd(M, K, scale=K ** -0.5) for _ in range(count)]

    TM_SINGLE = [("128-row tiles (default)", {}), ("256 x 128 x 2 stages", {"GGML_MI355X_GEMM_RING_TM256": 1282, "GGML_MI355X_GEMM_RING_TM256_MIN": 1}),
                 ("256 x 128 x 3 stages", {"GGML_MI355X_GEMM_RING_TM256": 1283, "GGML_MI355X_GEMM_RING_TM256_MIN": 1}),
                 ("256 x 64 x 2 stages", {"GGML_MI355X_GEMM_RING_TM256": 642, "GGML_MI355X_GEMM_RING_TM256_MIN": 1}),
                 ("256 x 64 x 3 stages", {"GGML_MI355X_GEMM_RING_TM256": 643, "GGML_MI355X_GEMM_RING_TM256_MIN": 1})]
    GROUP = [("128 x 64 x 2 (default)", {}), ("128 x 128 x 2", {"GGML_MI355X_GEMM_GROUP_CFG": 1282}),
             ("256 x 128 x 2", {"GGML_MI355X_GEMM_GROUP_CFG": 2561282}), ("256 x 128 x 3", {"GGML_MI355X_GEMM_GROUP_CFG": 2561283}),
             ("256 x 64 x 2", {"GGML_MI355X_GEMM_GROUP_CFG": 256642}), ("256 x 64 x 3", {"GGML_MI355X_GEMM_GROUP_CFG": 256643})]

    # ---- fc1: 1280 -> 5120, bias + GELU, epilogue writes fc2's prepared activations only
    if "fc1" in what:
        ws = pool(4 * n, n, 24)
        ep = ka.Epilogue()
        ep.bias, ep.gelu = bias.data_ptr(), 1

        def fc1(i):
            tw = ka.tensor(ws[i % len(ws)].data_ptr(), ka.F16, [n, 4 * n])
            rc = L.mi355x_gemm_f16act_prep(ctx.h, C.byref(tw), act.data_ptr(), n, T, None, 0, C.byref(ep), prep.data_ptr())
            return rc or L.mi355x_flush(ctx.h)
        for name, e in TM_SINGLE:
            timed("fc1", name, e, fc1, 2.0 * T * n * 4 * n, a.iters)
        del ws

    # ---- fc1 again, by epilogue: what the f16-table GELU and the Q8_0 rounding for fc2 cost on top of the product
    if "fc1ep" in what:
        ws = pool(4 * n, n, 24)
        for name, gelu, mode in (("no epilogue, F32 store", 0, "plain0"), ("bias, F32 store", 0, "plain"), ("bias + GELU, F32 store", 1, "plain"),
                                 ("bias + GELU, F32 + prepared", 1, "both"), ("bias + GELU, prepared only", 1, "only"), ("bias, prepared only", 0, "only")):
            epx = ka.Epilogue()
            epx.bias, epx.gelu = bias.data_ptr(), gelu

            def fc1x(i, epx=epx, mode=mode):
                tw = ka.tensor(ws[i % len(ws)].data_ptr(), ka.F16, [n, 4 * n])
                if mode == "plain0":
                    rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), n, T, y.data_ptr(), 4 * n * 4, ka.F32, None)
                elif mode == "plain":
                    rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), n, T, y.data_ptr(), 4 * n * 4, ka.F32, C.byref(epx))
                else:
                    rc = L.mi355x_gemm_f16act_prep(ctx.h, C.byref(tw), act.data_ptr(), n, T, y.data_ptr() if mode == "both" else None, 4 * n * 4 if mode == "both" else 0,
                                                   C.byref(epx), prep.data_ptr())
                return rc or L.mi355x_flush(ctx.h)
            timed("fc1", name, {}, fc1x, 2.0 * T * n * 4 * n, a.iters)
        del ws

    # ---- fc2: 5120 -> 1280, bias + residual
    if "fc2" in what:
        ws = pool(n, 4 * n, 24)
        ep2 = ka.Epilogue()
        ep2.bias, ep2.residual, ep2.residual_nb1 = bias.data_ptr(), res.data_ptr(), n * 4

        def fc2(i):
            tw = ka.tensor(ws[i % len(ws)].data_ptr(), ka.F16, [4 * n, n])
            rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act4.data_ptr(), 4 * n, T, ys[0].data_ptr(), n * 4, ka.F32, C.byref(ep2))
            return rc or L.mi355x_flush(ctx.h)
        for name, e in TM_SINGLE[:1] + TM_SINGLE[3:]:
            timed("fc2", name, e, fc2, 2.0 * T * n * 4 * n, a.iters)
        del ws

    # ---- O-projection: 1280 -> 1280, bias + residual
    if "oproj" in what:
        ws = pool(n, n, 96)
        ep3 = ka.Epilogue()
        ep3.bias, ep3.residual, ep3.residual_nb1 = bias.data_ptr(), res.data_ptr(), n * 4

        def oproj(i):
            tw = ka.tensor(ws[i % len(ws)].data_ptr(), ka.F16, [n, n])
            rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), n, T, ys[0].data_ptr(), n * 4, ka.F32, C.byref(ep3))
            return rc or L.mi355x_flush(ctx.h)
        for name, e in TM_SINGLE[:1] + TM_SINGLE[3:]:
            timed("oproj", name, e, oproj, 2.0 * T * n * n, a.iters)

    # ---- grouped launches: Q / K / V of a layer (3 members) and the cross-attention K / V of four decoder layers (8 members)
    for label, members in (("qkv", 3), ("xkv", 8)):
        if label not in what:
            continue
        ws = pool(n, n, 96)
        epb = ka.Epilogue()
        epb.bias = bias.data_ptr()

        def group(i, members=members, ws=ws):
            for j in range(members):
                tw = ka.tensor(ws[(i * members + j) % len(ws)].data_ptr(), ka.F16, [n, n])
                rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), n, T, ys[j].data_ptr(), n * 4, ka.F32, C.byref(epb))
                if rc:
                    return rc
            return L.mi355x_flush(ctx.h)
        for name, e in GROUP:
            timed(label, name, e, group, members * 2.0 * T * n * n, a.iters)
    ctx.close()


if __name__ == "__main__":
    main()
