#!/usr/bin/env python3
"""How the ENCODER work of concurrent streams sits beside the merged decode chains (rocprofv3 kernel trace of an S-stream run):
kernels are classed as `enc` (the wide GEMMs, the MFMA attention, LayerNorm / im2col / conv of a 1500-column graph: everything launched between a k_im2col_1d and
the first decode kernel of the same queue is not needed — the class is by kernel name and grid) or `chain` (merged-chain kernels: k_gemv_mx*, k_fattn_dec_multi, k_act_prepare,
k_fattn_self_q, k_decode_head_multi, k_gemv_q, k_vocab*), and the timeline is cut at every start / end:
  * share of wall time with only encoder kernels / only chain kernels / both / neither on the device;
  * mean duration of the big chain kernels when they run alone and when an encoder kernel overlaps them, and vice versa.
   usage: encode_overlap.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

ENC = ("k_gemm_f16_ring", "k_gemm_dq", "k_mmq", "k_fattn_mfma", "k_im2col_1d", "k_transpose_f32", "k_prep_act", "k_gemm_mfma", "k_dequant_f16")
CHAIN = ("k_gemv_mx", "k_fattn_dec_multi", "k_act_prepare", "k_fattn_self_q", "k_decode_head_multi", "k_gemv_q", "k_vocab", "k_gemv8", "k_scatter_upload", "k_gemv_row", "k_fattn_dec<")

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            g = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
            cls = "enc" if n.startswith(ENC) or (n.startswith("k_norm_v4") and g >= 90000) else ("chain" if n.startswith(CHAIN) else "other")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, cls))
rows.sort()
if not rows:
    print("no kernel trace"); sys.exit(1)
# skip the warm-up: analyse the second half of the trace (the timed chunks)
t_mid = rows[0][0] + (rows[-1][1] - rows[0][0]) // 2
rows = [r for r in rows if r[0] >= t_mid]
ev = []
for s, e, n, c in rows:
    ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
act = defaultdict(int)
share = defaultdict(int)
prev = ev[0][0]
for t, d, c in ev:
    if t > prev:
        k = ("enc" if act["enc"] else "") + ("+" if act["enc"] and act["chain"] else "") + ("chain" if act["chain"] else "")
        share[k or ("other" if act["other"] else "idle")] += t - prev
        prev = t
    act[c] += d
tot = sum(share.values())
print(f"analysed window: {tot / 1e6:.1f} ms, {len(rows)} kernels")
for k in ("chain", "enc", "enc+chain", "other", "idle"):
    print(f"  {k:10s} {share[k] / 1e6:9.2f} ms  {100 * share[k] / tot:5.1f} %")
# durations alone vs overlapped by the other class
import bisect
for cls, other in (("chain", "enc"), ("enc", "chain")):
    iv = sorted((s, e) for s, e, n, c in rows if c == other)
    starts = [i[0] for i in iv]
    agg = defaultdict(lambda: [0, 0.0, 0, 0.0])
    for s, e, n, c in rows:
        if c != cls:
            continue
        i = max(0, bisect.bisect_left(starts, s) - 8)
        ov = any(a < e and b > s for a, b in iv[i:i + 64])
        a = agg[n]
        if ov:
            a[2] += 1; a[3] += e - s
        else:
            a[0] += 1; a[1] += e - s
    print(f"\n{cls} kernels: mean us alone | mean us while a(n) {other} kernel is on the device")
    for n in sorted(agg, key=lambda k: -(agg[k][1] + agg[k][3]))[:8]:
        a = agg[n]
        print(f"  {n[:60]:60s} alone n={a[0]:6d} {a[1] / max(a[0], 1) / 1e3:8.2f}   overlapped n={a[2]:6d} {a[3] / max(a[2], 1) / 1e3:8.2f}")
