#!/usr/bin/env python3
"""The vocabulary projection alone (LN + [51866 x 1280] mat-vec, T columns) through the C ABI, HBM-cold: `--copies` distinct weight
matrices (default 8 x 45.6 MB > the 256 MB Infinity Cache) are visited round-robin, as in a decode step where 0.5 GB of other weights
pass between two visits.  Prints the hipEvent-bracketed per-launch average and the GB/s it means.  Knobs are environment variables the
library reads once (GGML_MI355X_GEMV_PASS_WAVES, GGML_MI355X_GEMV_WPB, GGML_MI355X_LOGITS_*): one process per setting.
  python scripts/logits_bench.py [--qtype q5_0] [--T 1] [--iters 64] [--copies 8]"""
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

ap = argparse.ArgumentParser()
ap.add_argument("--qtype", default="q5_0")
ap.add_argument("--T", type=int, default=1)
ap.add_argument("--iters", type=int, default=64)
ap.add_argument("--copies", type=int, default=8)
ap.add_argument("--N", type=int, default=51866)
ap.add_argument("--K", type=int, default=1280)
a = ap.parse_args()
import torch  # noqa: E402

tid = ka.TYPE_NAMES[a.qtype]
ctx = ka.Ctx(0)
L = ka.lib()
g = torch.Generator(device="cuda:0").manual_seed(0)
N, K, T = a.N, a.K, a.T
nbytes = N * ka.row_bytes(tid, K)
ws = []
for _ in range(a.copies):
    w = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
    nblk = N * K // 32
    w[nbytes - nblk * 2:] = (torch.rand(nblk, device="cuda:0", generator=g) * 0.01).half().view(torch.uint8)
    ws.append(w)
x = torch.randn((T, K), device="cuda:0", generator=g)
lw, lb = torch.ones(K, device="cuda:0"), torch.zeros(K, device="cuda:0")
y = torch.zeros((T, N), device="cuda:0")
descs = []
for w in ws:
    d = ka.GemvDesc()
    d.x, d.x_nb1, d.K, d.T, d.has_norm, d.eps = x.data_ptr(), K * 4, K, T, 1, 1e-5
    d.ln_w, d.ln_b, d.nseg = lw.data_ptr(), lb.data_ptr(), 1
    d.seg[0].w, d.seg[0].wtype, d.seg[0].N = w.data_ptr(), tid, N
    d.seg[0].dst, d.seg[0].dst_type, d.seg[0].dst_nb1 = y.data_ptr(), ka.F32, N * 4
    descs.append(d)
i = 0


def launch():
    global i
    rc = L.mi355x_gemv_fused(ctx.h, C.byref(descs[i % len(descs)]))
    i += 1
    assert rc == 0, (rc, L.mi355x_last_error())


for _ in range(2 * a.copies):
    launch()
ctx.sync()
ctx.prof(True)
ctx.prof_reset()
for _ in range(a.iters):
    launch()
rows = ctx.prof_report()
ctx.prof(False)
tot = sum(r["total_ms"] for r in rows)
us = tot * 1e3 / a.iters
print(json.dumps({"qtype": a.qtype, "T": T, "N": N, "K": K, "copies": a.copies, "us_per_launch": round(us, 2), "weight_MB": round(nbytes / 1e6, 2),
                  "GBps": round(nbytes / (us * 1e-6) / 1e9, 1), "kernels": {r["name"]: round(r["total_ms"] * 1e3 / max(r["calls"], 1), 2) for r in rows},
                  "env": {k: v for k, v in os.environ.items() if k.startswith("GGML_MI355X_")}}))
ctx.close()
