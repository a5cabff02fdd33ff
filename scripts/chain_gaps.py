#!/usr/bin/env python3
"""Where the time BETWEEN the kernels of a merged chain goes (rocprofv3 kernel trace): for the queue with the most chain steps, the gap in front of
every kernel (its start minus the previous kernel's end on the same queue), grouped by the kernel that waited and by the kernel it waited for,
and how much of each gap was covered by kernels of OTHER queues (the other chain, encoders of other states).
   usage: chain_gaps.py <dir with *kernel_trace.csv>"""
import bisect
import csv
import glob
import os
import statistics as st
import sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "0")))
rows.sort()
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
chainq = max(byq, key=lambda q: sum(1 for r in byq[q] if r[2].startswith("k_decode_head_multi")))
rs = byq[chainq]
others = sorted((r[0], r[1]) for r in rows if r[3] != chainq)
ostarts = [o[0] for o in others]


def covered(a, b):
    """ns of [a, b) during which at least one kernel of another queue ran (the others are merged first)"""
    i = max(0, bisect.bisect_left(ostarts, a) - 64)
    tot, cur_s, cur_e = 0, None, None
    for s, e in others[i:]:
        if s >= b:
            break
        s, e = max(s, a), min(e, b)
        if e <= s:
            continue
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


first = next(i for i, r in enumerate(rs) if r[2].startswith("k_decode_head_multi"))
wait_for, waited, cov_all, gaps = defaultdict(list), defaultdict(list), [], []
for prev, cur in zip(rs[first:-1], rs[first + 1:]):
    g = cur[0] - prev[1]
    if g > 200_000:            # between steps of the host (sampling, graph build): not a launch gap
        continue
    g = max(g, 0)
    c = covered(prev[1], cur[0]) if g > 0 else 0
    gaps.append(g / 1e3)
    cov_all.append(c / 1e3)
    waited[cur[2][:48]].append(g / 1e3)
    wait_for[prev[2][:48]].append(g / 1e3)
print(f"queue {chainq}: {len(gaps)} launch gaps, mean {st.mean(gaps):.2f} us, median {st.median(gaps):.2f} us, p90 {sorted(gaps)[int(0.9 * len(gaps))]:.2f} us; "
      f"covered by other queues' kernels: {100 * sum(cov_all) / max(sum(gaps), 1e-9):.0f} % of the gap time")
print("histogram (us):", {f"<{hi}": sum(1 for g in gaps if lo <= g < hi) for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 5), (5, 10), (10, 30), (30, 200))})
for title, d in (("gap in FRONT of", waited), ("gap BEHIND", wait_for)):
    print(f"  {title:18s} {'kernel':48s} {'n':>7s} {'mean_us':>8s} {'p50_us':>8s} {'total_ms':>9s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"  {'':18s} {k:48s} {len(v):7d} {st.mean(v):8.2f} {st.median(v):8.2f} {sum(v) / 1e3:9.2f}")

# ---- where in a chain step the gaps sit, and one step in full ----
heads = [i for i, r in enumerate(rs) if r[2].startswith("k_decode_head_multi")]
starts = [i for n, i in enumerate(heads) if n == 0 or i - heads[n - 1] > 4]
bins = defaultdict(list)
for a, b in zip(starts[2:-1], starts[3:]):
    for j in range(a + 1, b):
        g = rs[j][0] - rs[j - 1][1]
        if 0 <= g < 200_000:
            bins[(j - a) // 40].append(g / 1e3)
print("  gap by position in the step (launch index / 40): " + "  ".join(f"{k * 40}:{st.mean(v):.2f}" for k, v in sorted(bins.items()) if len(v) > 50))
if len(starts) > 12:
    a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
    print(f"  one step ({b - a} launches): start_us dur_us gap_us covered_by_other_queues_us kernel")
    t0 = rs[a][0]
    for j in range(a, min(b, a + 140)):
        g = rs[j][0] - rs[j - 1][1]
        print(f"    {(rs[j][0] - t0) / 1e3:9.2f} {(rs[j][1] - rs[j][0]) / 1e3:7.2f} {g / 1e3:7.2f} {covered(rs[j - 1][1], rs[j][0]) / 1e3 if g > 0 else 0:7.2f} {rs[j][2][:60]}")
