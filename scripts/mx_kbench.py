#!/usr/bin/env python3
"""Micro benchmark of the plane mat-vecs of a merged decode chain (cross-state batch) through the C ABI: the six products of one
large-v3 decoder layer + the vocabulary projection over PREPARED activation planes, T = 8 .. 32 columns, HBM-cold weights (every
launch reads another copy; the copies of a case total ~0.5 GB > the 256 MB Infinity Cache, as in a real step, whose 550 MB of
weights are touched once).  hipEvent-bracketed per-launch averages from the library's own profiler.

  GGML_MI355X_MX_MIN_T=9  python scripts/mx_kbench.py      # matrix-core form (decode_mx.hip) from 9 columns
  GGML_MI355X_MX_MIN_T=0  python scripts/mx_kbench.py      # k_gemv_q / k_vocab (decode_q.hip)
"""
import argparse
import ctypes as C
import json
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
    ap.add_argument("--qtype", default="q5_0")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--T", default="8,12,16,24,32")
    ap.add_argument("--vocab", type=int, default=1)
    ap.add_argument("--cold-mb", type=int, default=500)
    a = ap.parse_args()
    import torch
    tid = ka.TYPE_NAMES[a.qtype]
    ctx = ka.Ctx(0)
    L = ka.lib()
    n = 1280
    g = torch.Generator(device="cuda:0").manual_seed(0)

    def wq(N, K):
        nbytes = N * ka.row_bytes(tid, K)
        w = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
        nblk = N * K // 32
        w[nbytes - nblk * 2:] = (torch.rand(nblk, device="cuda:0", generator=g) * 0.01).half().view(torch.uint8)
        return w

    shapes = [("qkv 1280->3x1280", n, [n, n, n], 0, False), ("oproj 1280->1280 +res", n, [n], 0, True), ("fc1 1280->5120 gelu ->planes", n, [4 * n], 1, False),
              ("fc2 5120->1280 +res", 4 * n, [n], 0, True)]
    if a.vocab:
        shapes.append(("logits 1280->51866", n, [51866], 0, False))
    out = []
    for T in [int(t) for t in a.T.split(",")]:
        for name, K, segsN, gelu, resid in shapes:
            per = sum(N * ka.row_bytes(tid, K) for N in segsN)
            copies = max(2, min(256, (a.cold_mb << 20) // per))
            ws = [[wq(N, K) for N in segsN] for _ in range(copies)]
            x = torch.randn((T, K), device="cuda:0", generator=g)
            ys = [torch.zeros((T, N), device="cuda:0") for N in segsN]
            bias = torch.zeros(max(segsN), device="cuda:0")
            p0, p1 = L.mi355x_act_scratch(ctx.h, 0), L.mi355x_act_scratch(ctx.h, 1)
            ad = ka.ActDesc()
            ad.x, ad.x_nb1, ad.K, ad.T, ad.wtype = x.data_ptr(), K * 4, K, T, tid
            ctx.check(L.mi355x_act_prepare(ctx.h, C.byref(ad), p0), "act_prepare")
            descs = []
            for wset in ws:
                d = ka.GemvDesc()
                d.K, d.T, d.nseg, d.x_planes = K, T, len(segsN), p0
                for s, N in enumerate(segsN):
                    d.seg[s].w, d.seg[s].wtype, d.seg[s].N = wset[s].data_ptr(), tid, N
                    d.seg[s].ep = ka.Epilogue(bias.data_ptr() if N <= 8192 else None, 0.0, 0, gelu, ys[s].data_ptr() if resid else None, N * 4)
                    d.seg[s].dst, d.seg[s].dst_type, d.seg[s].dst_nb1 = ys[s].data_ptr(), ka.F32, N * 4
                if gelu:
                    d.planes_out, d.planes_out_only = p1, 1
                descs.append(d)
            state = {"i": 0}

            def fn():
                d = descs[state["i"] % len(descs)]
                state["i"] += 1
                return L.mi355x_gemv_fused(ctx.h, C.byref(d))
            rc = 0
            for _ in range(max(8, copies)):
                rc = fn()
                if rc:
                    break
            ctx.sync()
            if rc:
                out.append({"case": name, "T": T, "error": f"rc={rc} {L.mi355x_last_error()}"})
                continue
            ctx.prof(True)
            ctx.prof_reset()
            for _ in range(a.iters):
                fn()
            rows = ctx.prof_report()
            ctx.prof(False)
            tot = sum(r["total_ms"] for r in rows)
            out.append({"case": name, "T": T, "us": round(tot * 1e3 / a.iters, 2), "MB": round(per / 1e6, 2), "copies": copies,
                        "kernels": {r["name"]: round(r["total_ms"] * 1e3 / max(r["calls"], 1), 2) for r in rows}})
            del ws, descs
            torch.cuda.empty_cache()
    print(json.dumps({"qtype": a.qtype, "mx_min_t": os.environ.get("GGML_MI355X_MX_MIN_T", "default"), "cases": out}, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
