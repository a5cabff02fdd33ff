"""Summarise a rocprofv3 --kernel-trace of scripts/stall_probe: mean start-gap before each kernel of the 5-kernel burst."""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = [r for r in rows if "k_spin" not in r[2]]
gaps = defaultdict(list)
for prev, cur in zip(rows, rows[1:]):
    m = re.search(r"k_(small|wide)<(\d+)>", cur[2])
    if not m: continue
    tag = int(m.group(2)); slot = {("small", 0): None, ("wide", 0): "s->W1", ("wide", 1): "W1->W2", ("small", 1): "W2->s2", ("wide", 2): "s2->W3"}[(m.group(1), tag // 1000)]
    if slot: gaps[(tag % 1000, slot)].append((cur[0] - prev[1]) / 1e3)
print("idle_us  " + "  ".join(f"{s:>8}" for s in ("s->W1", "W1->W2", "W2->s2", "s2->W3")))
for idle in sorted({k[0] for k in gaps}):
    print(f"{idle:7d}  " + "  ".join(f"{sum(gaps[(idle, s)]) / max(1, len(gaps[(idle, s)])):8.2f}" for s in ("s->W1", "W1->W2", "W2->s2", "s2->W3")))
