#!/usr/bin/env python3
"""Timeline of merged-chain steps per queue from a rocprofv3 kernel trace: a step = the kernels of one queue from a k_decode_head_multi up to the next one;
prints, for a window in the second half of the trace, every step (queue, first kernel start, last kernel end, busy time inside, kernels) in start order, the device-idle
gaps (no kernel of any queue running) longer than 50 us with what ended before and what started after them, and a histogram of idle-gap lengths.
   usage: chain_timeline.py <dir with *kernel_trace.csv> [n_steps=40]"""
import csv, glob, os, sys
from collections import defaultdict
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "0")))
rows.sort()
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * 6 // 10
rows = [r for r in rows if r[0] >= t0]
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
steps = []
for q, rs in byq.items():
    cur = None
    for s, e, n, _ in rs:
        if n.startswith("k_decode_head_multi"):
            if cur: steps.append(cur)
            cur = {"q": q, "s": s, "e": e, "busy": 0, "n": 0, "cols": None}
        if cur is None:
            continue
        cur["e"] = max(cur["e"], e); cur["busy"] += e - s; cur["n"] += 1
    if cur: steps.append(cur)
steps.sort(key=lambda x: x["s"])
base = steps[0]["s"] if steps else t0
print("chain steps (ms from the window's first step): queue  start  end  span  kernel-busy  kernels")
for st in steps[:nshow]:
    print(f"  q{st['q']:>3s}  {(st['s'] - base) / 1e6:8.3f} {(st['e'] - base) / 1e6:8.3f}  span {(st['e'] - st['s']) / 1e6:6.3f}  busy {st['busy'] / 1e6:6.3f}  n={st['n']}")
# idle gaps over all queues
ev = sorted(rows, key=lambda r: r[0])
end = ev[0][1]; last_name = ev[0][2]; gaps = []
for s, e, n, q in ev[1:]:
    if s > end:
        gaps.append((s - end, end, last_name, n, q))
    if e > end:
        end = e; last_name = n
tot = ev[-1][1] - ev[0][0]
idle = sum(g[0] for g in gaps)
print(f"\nwindow {tot / 1e6:.1f} ms, device idle {idle / 1e6:.1f} ms = {100 * idle / tot:.1f} %")
bins = [(0, 5), (5, 20), (20, 50), (50, 200), (200, 500), (500, 1000), (1000, 1e9)]
for lo, hi in bins:
    g = [x[0] for x in gaps if lo * 1e3 <= x[0] < hi * 1e3]
    print(f"  gaps {lo:>5}-{hi if hi < 1e8 else 'inf':>5} us: n={len(g):6d}  total {sum(g) / 1e6:8.2f} ms")
after = defaultdict(lambda: [0, 0]); before = defaultdict(lambda: [0, 0])
for d, t, ln, nn, q in gaps:
    if d >= 50e3:
        after[nn][0] += 1; after[nn][1] += d; before[ln][0] += 1; before[ln][1] += d
print("gaps >= 50 us by the kernel that ENDED them (first kernel after the idle time):")
for k in sorted(after, key=lambda k: -after[k][1])[:6]: print(f"  {k[:60]:60s} n={after[k][0]:5d} total {after[k][1] / 1e6:8.2f} ms")
print("... and by the last kernel BEFORE them:")
for k in sorted(before, key=lambda k: -before[k][1])[:6]: print(f"  {k[:60]:60s} n={before[k][0]:5d} total {before[k][1] / 1e6:8.2f} ms")
