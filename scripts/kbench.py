#!/usr/bin/env python3
"""Kernel-level micro benchmark through the C ABI (include/mi355x_kernels.h): the decode-step shapes of one
large-v3 layer, each launched `--iters` times back to back on the library's stream.  Meant to run under
`rocprofv3 --kernel-trace -f csv` (scripts/summarize_trace.py then reports per-(kernel, grid) durations); it also prints
the hipEvent-bracketed per-launch averages measured by the library's own profiler.

  python scripts/kbench.py [--qtype q5_0] [--iters 50] [--T 1]"""
import argparse
import ctypes as C
import json
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
    ap.add_argument("--qtype", default="q5_0")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--T", type=int, default=1)
    ap.add_argument("--cycle", action="store_true", help="add the cache-level experiment (needs ~0.6 GB)")
    a = ap.parse_args()
    import torch
    tid = ka.TYPE_NAMES[a.qtype]
    ctx = ka.Ctx(0)
    L = ka.lib()
    T, n = a.T, 1280
    g = torch.Generator(device="cuda:0").manual_seed(0)

    def wq(N, K):   # random bytes are valid blocks for timing purposes (d planes: small finite halves)
        nbytes = N * ka.row_bytes(tid, K)
        w = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
        nblk = N * K // 32
        dpl = (torch.rand(nblk, device="cuda:0", generator=g) * 0.01).half().view(torch.uint8)
        w[nbytes - nblk * 2:] = dpl          # d plane is the last plane of every layout
        return w

    x = torch.randn((T, n), device="cuda:0", generator=g)
    x4 = torch.randn((T, 4 * n), device="cuda:0", generator=g)
    lw, lb = torch.ones(n, device="cuda:0"), torch.zeros(n, device="cuda:0")
    bias = torch.zeros(4 * n, device="cuda:0")
    cases = {}

    def gemv_case(name, K, segsN, norm, xt, gelu=0, resid=False):
        ws = [wq(N, K) for N in segsN]
        ys = [torch.zeros((T, N), device="cuda:0") for N in segsN]
        d = ka.GemvDesc()
        d.x, d.x_nb1, d.K, d.T, d.has_norm, d.eps = xt.data_ptr(), K * 4, K, T, 1 if norm else 0, 1e-5
        d.ln_w, d.ln_b, d.nseg = lw.data_ptr(), lb.data_ptr(), len(segsN)
        for s, N in enumerate(segsN):
            d.seg[s].w, d.seg[s].wtype, d.seg[s].N = ws[s].data_ptr(), tid, N
            d.seg[s].ep = ka.Epilogue(bias.data_ptr(), 0.0, 0, gelu, ys[s].data_ptr() if resid else None, N * 4)
            d.seg[s].dst, d.seg[s].dst_type, d.seg[s].dst_nb1 = ys[s].data_ptr(), ka.F32, N * 4
        cases[name] = (lambda: L.mi355x_gemv_fused(ctx.h, C.byref(d)), (ws, ys, d), sum(N * ka.row_bytes(tid, K) for N in segsN))

    gemv_case("ln+qkv 1280->3x1280", n, [n, n, n], True, x)
    gemv_case("oproj 1280->1280 +res", n, [n], False, x, resid=True)
    gemv_case("ln+fc1 1280->5120 gelu", n, [4 * n], True, x, gelu=1)
    gemv_case("fc2 5120->1280 +res", 4 * n, [n], False, x4, resid=True)
    gemv_case("ln+logits 1280->51866", n, [51866], True, x)

    # cache-level experiment: the same O-projection over `copies` distinct weight matrices visited round-robin
    # (1: L2-hot, 16: 18 MB = L2-resident, 64: 72 MB = Infinity-Cache-resident, 400: 450 MB = HBM-cold)
    if a.cycle:
        for copies in (1, 16, 64, 400):
            ws = [wq(n, n) for _ in range(copies)]
            y = torch.zeros((T, n), device="cuda:0")
            descs = []
            for w_ in ws:
                d = ka.GemvDesc()
                d.x, d.x_nb1, d.K, d.T, d.has_norm, d.eps = x.data_ptr(), n * 4, n, T, 0, 1e-5
                d.nseg = 1
                d.seg[0].w, d.seg[0].wtype, d.seg[0].N = w_.data_ptr(), tid, n
                d.seg[0].dst, d.seg[0].dst_type, d.seg[0].dst_nb1 = y.data_ptr(), ka.F32, n * 4
                descs.append(d)
            state = {"i": 0}

            def cyc(descs=descs, state=state):
                d = descs[state["i"] % len(descs)]
                state["i"] += 1
                return L.mi355x_gemv_fused(ctx.h, C.byref(d))
            cases[f"oproj cycle x{copies}"] = (cyc, (ws, y, descs), n * ka.row_bytes(tid, n))

    # producer -> consumer: the activation vector is (re)written by a preceding kernel on the same stream, as in the decode graph
    if a.cycle:
        wp = wq(n, n)
        x2 = torch.zeros((T, n), device="cuda:0")
        yp = torch.zeros((T, n), device="cuda:0")
        tx1, tx2 = ka.tensor(x.data_ptr(), ka.F32, [n, T]), ka.tensor(x2.data_ptr(), ka.F32, [n, T])
        dp = ka.GemvDesc()
        dp.x, dp.x_nb1, dp.K, dp.T, dp.has_norm, dp.eps, dp.nseg = x2.data_ptr(), n * 4, n, T, 0, 1e-5, 1
        dp.seg[0].w, dp.seg[0].wtype, dp.seg[0].N = wp.data_ptr(), tid, n
        dp.seg[0].dst, dp.seg[0].dst_type, dp.seg[0].dst_nb1 = yp.data_ptr(), ka.F32, n * 4

        def prodcons():
            rc = L.mi355x_scale(ctx.h, C.byref(tx1), C.byref(tx2), 1.0001, 0.0)
            return rc or L.mi355x_gemv_fused(ctx.h, C.byref(dp))
        cases["scale -> oproj (fresh activations)"] = (prodcons, (wp, x2, yp, dp), n * ka.row_bytes(tid, n))

    H, D = 20, 64
    for nm, n_kv in (("xattn kv1536", 1536), ("self kv64", 64), ("self kv300", 300)):
        q = torch.randn((T, H, D), device="cuda:0", generator=g)
        k = torch.randn((n_kv, H, D), device="cuda:0", generator=g).half()
        v = torch.randn((n_kv, H, D), device="cuda:0", generator=g).half()
        o = torch.zeros((T, H, D), device="cuda:0")
        tq = ka.tensor(q.data_ptr(), ka.F32, [D, T, H], [4, H * D * 4, D * 4, T * H * D * 4])
        tk = ka.tensor(k.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
        tv = ka.tensor(v.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
        to = ka.tensor(o.data_ptr(), ka.F32, [D, H, T])
        cases[nm] = (lambda tq=tq, tk=tk, tv=tv, to=to: L.mi355x_flash_attn_ext(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), None, C.byref(to), 0.125),
                     (q, k, v, o, tq, tk, tv, to), 2 * n_kv * H * D * 2)
        # partials + projection with the combine in its prologue
        wo_ = wq(n, n)
        yo = torch.zeros((T, n), device="cuda:0")

        def fused(tq=tq, tk=tk, tv=tv, wo_=wo_, yo=yo):
            parts = ka.AttnPartials()
            rc = L.mi355x_flash_attn_partial(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), None, 0.125, C.byref(parts))
            if rc:
                return rc
            d = ka.GemvDesc()
            d.K, d.T, d.nseg = n, T, 1
            d.attn_part_o, d.attn_part_ml, d.attn_nparts = parts.part_o, parts.part_ml, parts.nparts
            d.seg[0].w, d.seg[0].wtype, d.seg[0].N = wo_.data_ptr(), tid, n
            d.seg[0].dst, d.seg[0].dst_type, d.seg[0].dst_nb1 = yo.data_ptr(), ka.F32, n * 4
            return L.mi355x_gemv_fused(ctx.h, C.byref(d))
        cases[nm + " + oproj(fused combine)"] = (fused, (wo_, yo), 2 * n_kv * H * D * 2 + n * ka.row_bytes(tid, n))

    torch.cuda.synchronize()
    out = []
    for name, (fn, keep, nbytes) in cases.items():
        for _ in range(3):
            rc = fn()
            if rc:
                print(f"{name}: rc={rc} {L.mi355x_last_error()}", file=sys.stderr)
                break
        ctx.sync()
        if rc:
            continue
        st = (C.c_uint64 * 16)()
        if L.mi355x_debug_read_stamps(ctx.h, st) == 0 and st[6] > st[0] > 0:
            print(f"anatomy {name}: " + " ".join(f"{st[i + 1] - st[i]}" for i in range(6)) + f"  total={st[6] - st[0]} ticks (s_memtime)", file=sys.stderr)
        iters = a.iters if "cycle" not in name else max(a.iters, 2 * int(name.split("x")[-1]))
        for _ in range(iters if "cycle" in name else 0):      # one full pass first: page-table / first-touch effects out of the way
            fn()
        ctx.sync()
        ctx.prof(True)
        ctx.prof_reset()
        for _ in range(iters):
            fn()
        rows = ctx.prof_report()
        ctx.prof(False)
        tot = sum(r["total_ms"] for r in rows)
        out.append({"case": name, "us_per_call_events": round(tot * 1e3 / iters, 2), "algo_MB": round(nbytes / 1e6, 3),
                    "GBps": round(nbytes / (tot * 1e-3 / iters) / 1e9, 1) if tot else None,
                    "kernels": {r["name"]: round(r["total_ms"] * 1e3 / max(r["calls"], 1), 2) for r in rows}})
        # the same launches without event bracketing (what rocprofv3 sees as back-to-back dispatches)
        for _ in range(iters):
            fn()
        ctx.sync()
    print(json.dumps({"qtype": a.qtype, "T": T, "iters": a.iters, "cases": out}, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
