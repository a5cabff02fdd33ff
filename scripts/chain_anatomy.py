#!/usr/bin/env python3
"""Merged-chain anatomy from a rocprofv3 kernel trace: a chain step = the kernels from one k_decode_head_multi (token embedding of all
columns) to the next on the same queue.  Prints per-kernel totals and, per step, GPU busy time against the step's span.
   usage: chain_anatomy.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import statistics as st
import sys
from collections import defaultdict

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "0"), r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
rows.sort()
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
for q, rs in byq.items():
    idx = [i for i, r in enumerate(rs) if r[2].startswith("k_decode_head_multi") and int(r[4]) > 0]
    # a step has two head launches (embedding, mask cast); steps start at the first of a pair
    starts = [i for n, i in enumerate(idx) if n == 0 or i - idx[n - 1] > 4]
    if len(starts) < 8:
        continue
    spans, busys, counts = [], [], []
    agg = defaultdict(list)
    for a, b in zip(starts[2:-1], starts[3:]):
        seg = rs[a:b]
        spans.append((seg[-1][1] - seg[0][0]) / 1e3)
        busys.append(sum(e - s for s, e, *_ in seg) / 1e3)
        counts.append(len(seg))
        for s, e, name, _, g, wg in seg:
            agg[(name[:52], g, wg)].append((e - s) / 1e3)
    nsteps = len(spans)
    print(f"queue {q}: {nsteps} chain steps, launches/step median {st.median(counts)}, GPU busy/step median {st.median(busys):.1f} us, span/step median {st.median(spans):.1f} us "
          f"(first kernel start -> last kernel end), step period median {st.median([(rs[b][0] - rs[a][0]) / 1e3 for a, b in zip(starts[2:-1], starts[3:])]):.1f} us")
    print(f"  {'kernel':52s} {'grid':>7s} {'wg':>5s} {'n/step':>7s} {'avg_us':>8s} {'us/step':>9s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
        print(f"  {k[0]:52s} {k[1]:>7s} {k[2]:>5s} {len(v)/nsteps:7.1f} {sum(v)/len(v):8.2f} {sum(v)/nsteps:9.1f}")
