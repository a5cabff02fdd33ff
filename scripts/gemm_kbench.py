#!/usr/bin/env python3
"""Encoder-GEMM micro benchmark through the C ABI (include/mi355x_kernels.h): the wide products of one large-v3 encoder layer
(Q/K/V group, O projection, fc1 + GELU -> fc2's activations, fc2 + residual, the cross-K/V group of 8) on the three GEMM families:

  int8   mi355x_gemm_q8act       k_mmq: int8 MFMA on the quantized operands (exact integer sums, r04)
  dq     mi355x_gemm_f16act      k_gemm_dq: quantized A unpacked per workgroup into LDS, f16 MFMA (r06); token-tile width forced to 128 / 256 or chosen
  ring   mi355x_gemm_f16act      k_gemm_f16_ring: the weight's f16 copy through the LDS-DMA ring (r02-r03, GGML_MI355X_MMQ=0)

Per case: hipEvent-bracketed mean per launch from the library's profiler, TFLOP/s (TOP/s), and whether dq == ring bit for bit.
  python scripts/gemm_kbench.py [--qtype q5_0] [--iters 30] [--cases qkv,o,fc1,fc2,xkv]"""
import argparse
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from whisper_cpp_amd import kernels_api as ka  # noqa: E402

OPT_DQ_GEMM, OPT_DQ_BN = 4, 5        # mi355x_kernels.h: enum after MMQ_GROUP, MMQ_SCALE_MFMA, MMQ_TILE, FATTN_NG


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qtype", default="q5_0")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--cases", default="qkv,o,fc1,fc2,xkv")
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--families", default="int8,dq,dq128,dq256,ring")
    ap.add_argument("--ldpad", type=int, default=0, help="pad the f16 activation rows by this many elements (row stride K + ldpad): L2-channel experiment")
    ap.add_argument("--ablate", default="", help="comma list of MI355X_OPT_DQ_ABLATE masks: times the dq family once per mask (results are garbage)")
    a = ap.parse_args()
    import torch
    tid = ka.TYPE_NAMES[a.qtype]
    ctx = ka.Ctx(0)
    L = ka.lib()
    L.mi355x_test_option.argtypes = [C.c_int, C.c_int, C.c_int]
    T, n = a.T, 1280
    g = torch.Generator(device="cuda:0").manual_seed(0)

    def wq(N, K):       # random bytes are valid blocks for timing; scale planes: small finite halves
        nbytes = N * ka.row_bytes(tid, K)
        w = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
        if a.qtype == "q4_k":
            nsb = N * K // 256
            w[nbytes - nsb * 4:] = (torch.rand(nsb * 2, device="cuda:0", generator=g) * 0.01).half().view(torch.uint8)
        else:
            nblk = N * K // 32
            w[nbytes - nblk * 2:] = (torch.rand(nblk, device="cuda:0", generator=g) * 0.01).half().view(torch.uint8)
        return w

    # name: (members, M, K, gelu+prep, residual)
    shapes = {"qkv": (3, n, n, False, False), "o": (1, n, n, False, True), "fc1": (1, 4 * n, n, True, False), "fc2": (1, n, 4 * n, False, True), "xkv": (8, n, n, False, False)}
    out = []
    for name in a.cases.split(","):
        nm, M, K, gelu, resid = shapes[name]
        ws = [wq(M, K) for _ in range(nm)]
        x = torch.randn((T, K), device="cuda:0", generator=g)
        mode = 2 if tid == ka.Q4_K else 1
        act = torch.zeros((T, K), dtype=torch.float16, device="cuda:0")
        rows = torch.zeros(L.mi355x_act_rows_bytes(tid, K, T), dtype=torch.uint8, device="cuda:0")
        prep = torch.zeros((T, M), dtype=torch.float16, device="cuda:0")
        prep_rows = torch.zeros(L.mi355x_act_rows_bytes(tid, M, T), dtype=torch.uint8, device="cuda:0")
        bias = torch.randn(M, device="cuda:0", generator=g) * 0.02
        res = torch.randn((T, M), device="cuda:0", generator=g)
        torch.cuda.synchronize()
        ctx.check(L.mi355x_prep_act(ctx.h, x.data_ptr(), K * 4, 0, act.data_ptr(), K, T, mode), "prep f16")
        ld = K + a.ldpad
        if a.ldpad:
            ctx.sync()
            actp = torch.zeros((T, ld), dtype=torch.float16, device="cuda:0")
            actp[:, :K] = act
            act = actp
            torch.cuda.synchronize()
        ctx.check(L.mi355x_prep_act(ctx.h, x.data_ptr(), K * 4, 0, rows.data_ptr(), K, T, 4 if tid == ka.Q4_K else 3), "prep rows")
        shadows = []
        for w in ws:
            sh = torch.zeros((M, K), dtype=torch.float16, device="cuda:0")
            ctx.check(L.mi355x_dequant_f16(ctx.h, C.byref(ka.tensor(w.data_ptr(), tid, [K, M])), sh.data_ptr()), "dequant")
            shadows.append(sh)
        ctx.sync()
        ep = ka.Epilogue(bias.data_ptr(), 0.0, 0, 1 if gelu else 0, res.data_ptr() if resid else None, M * 4)
        flop = 2.0 * nm * M * K * T
        results = {}

        def run(family):
            ys = [torch.zeros((T, M), dtype=torch.float32, device="cuda:0") for _ in range(nm)]
            torch.cuda.synchronize()
            L.mi355x_test_option(OPT_DQ_BN, 0, 0)
            L.mi355x_test_option(OPT_DQ_GEMM, 1 if family.startswith("dq") else 0, 1 if family.startswith("dq") else 0)
            if family in ("dq128", "dq256"):
                L.mi355x_test_option(OPT_DQ_BN, int(family[2:]), 1)

            def once():
                for i in range(nm):
                    if family == "int8":
                        tw = ka.tensor(ws[i].data_ptr(), tid, [K, M])
                        if gelu and tid != ka.Q4_K:
                            rc = L.mi355x_gemm_q8act_prep(ctx.h, C.byref(tw), rows.data_ptr(), T, None, M * 4, C.byref(ep), prep_rows.data_ptr())
                        else:
                            rc = L.mi355x_gemm_q8act(ctx.h, C.byref(tw), rows.data_ptr(), T, ys[i].data_ptr(), M * 4, ka.F32, C.byref(ep))
                    else:
                        tw = ka.tensor(shadows[i].data_ptr(), ka.F16, [K, M]) if family == "ring" else ka.tensor(ws[i].data_ptr(), tid, [K, M])
                        if gelu:
                            rc = L.mi355x_gemm_f16act_prep(ctx.h, C.byref(tw), act.data_ptr(), ld, T, None, M * 4, C.byref(ep), prep.data_ptr())
                        else:
                            rc = L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), ld, T, ys[i].data_ptr(), M * 4, ka.F32, C.byref(ep))
                    ctx.check(rc, f"{name} {family}")
                ctx.check(L.mi355x_flush(ctx.h), "flush")
            once(); ctx.sync()
            ctx.prof(True); ctx.prof_reset()
            for _ in range(a.iters):
                once()
            ctx.sync()
            rep = ctx.prof_report()
            ctx.prof(False)
            L.mi355x_test_option(OPT_DQ_BN, 0, 0)
            st = (C.c_uint64 * 16)()
            if family.startswith("dq") and L.mi355x_debug_read_stamps(ctx.h, st) == 0 and st[3] > st[0]:
                for base, who in ((0, "wave0"), (8, "last wave")):
                    e = [int(st[base + i]) for i in range(8)]
                    print(json.dumps({"case": name, "anatomy": who, "ablate": int(os.environ.get("_ABL", "0")), "ticks": {"prologue": e[1] - e[0], "loop": e[2] - e[1], "epilogue": e[3] - e[2],
                          "in_loop_wait+barrier": e[4], "in_loop_staging": e[5], "in_loop_mfma": e[6]}}), flush=True)
            total_ms = sum(r["total_ms"] for r in rep)
            us = total_ms * 1e3 / a.iters
            keep = prep.clone() if gelu else torch.stack(ys).clone()
            return us, keep, sorted({r["name"] for r in rep})

        for fam in a.families.split(","):
            if fam == "ring" and nm > 1 and False:
                continue
            us, y, names = run(fam)
            results[fam] = (us, y)
            out.append({"case": name, "family": fam, "qtype": a.qtype, "members": nm, "M": M, "K": K, "T": T, "us_per_step": round(us, 2),
                        "TFLOPs": round(flop / us / 1e6, 1), "kernels": names})
            print(json.dumps(out[-1]), flush=True)
        for m in [int(v) for v in a.ablate.split(",") if v]:
            L.mi355x_test_option(6, m, 1)
            os.environ["_ABL"] = str(m)
            us, _, names = run("dq")
            os.environ["_ABL"] = "0"
            L.mi355x_test_option(6, 0, 0)
            print(json.dumps({"case": name, "family": "dq", "ablate": m, "us_per_step": round(us, 2), "kernels": names}), flush=True)
        if "ring" in results:
            for fam in ("dq", "dq128", "dq256"):
                if fam in results:
                    same = bool(torch.equal(results[fam][1].view(torch.int16 if gelu else torch.int32), results["ring"][1].view(torch.int16 if gelu else torch.int32)))
                    print(json.dumps({"case": name, "check": f"{fam} == ring bit for bit", "ok": same}), flush=True)
        if "int8" in results and "dq" in results and not gelu:
            ref, got = results["int8"][1].double(), results["dq"][1].double()
            print(json.dumps({"case": name, "check": "NMSE dq vs int8", "nmse": float(((ref - got) ** 2).sum() / (ref ** 2).sum())}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
