"""Per-(kernel, grid, workgroup) duration statistics from a rocprofv3 `--kernel-trace -f csv` kernel_trace.csv."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(d) else [d]
    if not files:
        print("no kernel_trace.csv under", d)
        return 1
    agg = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                key = (name, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
                agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':44s} {'grid':>14s} {'wg':>4s} {'lds':>6s} {'n':>7s} {'avg_us':>8s} {'min_us':>8s} {'p50_us':>8s} {'max_us':>8s} {'total_ms':>9s} {'share':>6s}")
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print(f"{key[0][:44]:44s} {key[1] + 'x' + key[2]:>14s} {key[3]:>4s} {key[4]:>6s} {len(v):7d} {sum(v)/len(v):8.2f} {v[0]:8.2f} {v[len(v)//2]:8.2f} {v[-1]:8.2f} {sum(v)/1e3:9.3f} {100*sum(v)/tot:5.1f}%")
    print(f"total kernel time: {tot/1e3:.3f} ms")
    per_step(files)
    return 0


def per_step(files, marker="k_get_rows<6>"):
    """decode-step anatomy: kernels between two token-embedding gathers form one decode graph; for the most common step
    length print, per position, the mean duration and the mean gap to the previous kernel's end"""
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size_X", "?")))
    rows.sort()
    steps, cur = [], None
    for r in rows:
        if r[2].startswith(marker):
            if cur:
                steps.append(cur)
            cur = []
        if cur is not None:
            cur.append(r)
    if not steps:
        return
    from collections import Counter
    L = Counter(len(s) for s in steps).most_common(1)[0][0]
    sel = [s for s in steps if len(s) == L and all(a[2] == b[2] and a[3] == b[3] for a, b in zip(s, steps[[len(x) for x in steps].index(L)]))]
    if len(sel) < 4:
        return
    print(f"\ndecode-step anatomy: {len(sel)} steps of {L} kernels (marker {marker})")
    span = sum(s[-1][1] - s[0][0] for s in sel) / len(sel) / 1e3
    busy = sum(sum(k[1] - k[0] for k in s) for s in sel) / len(sel) / 1e3
    print(f"mean span first-start..last-end = {span:.1f} us, kernel-busy = {busy:.1f} us, gaps = {span - busy:.1f} us")
    print(f"{'pos':>4s} {'kernel':40s} {'grid':>8s} {'dur_us':>8s} {'gap_us':>8s} {'gap_med':>8s}")
    big = []
    for i in range(L):
        dur = sum(s[i][1] - s[i][0] for s in sel) / len(sel) / 1e3
        gaps = sorted((s[i][0] - s[i - 1][1]) / 1e3 for s in sel) if i else [0.0]
        gap, med = sum(gaps) / len(gaps), gaps[len(gaps) // 2]
        if gap > 2.0:
            big.append((i, gap, med))
        if i < 30 or i >= L - 6:
            print(f"{i:4d} {sel[0][i][2][:40]:40s} {sel[0][i][3]:>8s} {dur:8.2f} {gap:8.2f} {med:8.2f}")
        elif i == 30:
            print("   ... (layers repeat)")
    print("positions with mean gap > 2 us (pos, mean, median): " + ", ".join(f"{i}: {g:.1f}/{m:.1f}" for i, g, m in big))


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
