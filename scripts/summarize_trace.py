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
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
