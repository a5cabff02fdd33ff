#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of scripts/gpu_round.sh pmc.
bytes = counter * 1024; FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section);
WRITE_SIZE uncalibrated.   usage: make_pmc_traffic.py <gpurun_out> <round> > profiles/pmc_traffic.json"""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"].split("(")[0].strip()]
            a[0] += 1
            a[1] += float(r.get("Counter_Value") or 0)
    return acc


out_dir, rnd = sys.argv[1], int(sys.argv[2])
rd = per_kernel(os.path.join(out_dir, "pmc_FETCH_SIZE"), "FETCH_SIZE")
wr = per_kernel(os.path.join(out_dir, "pmc_WRITE_SIZE"), "WRITE_SIZE")
kern = {}
for name, (n, tot) in sorted(rd.items(), key=lambda kv: -kv[1][1]):
    if name.startswith("__amd") or n == 0:
        continue
    r = tot / n * 1024 * 2
    wn, wt = wr.get(name, [0, 0.0])
    w = wt / wn * 1024 if wn else 0.0
    kern[name] = {"launches_sampled": n, "hbm_read_bytes_per_launch": round(r), "hbm_write_bytes_per_launch": round(w), "hbm_bytes_per_launch": round(r + w)}
# optional: argv[3] = "<arch> <qtype>" of the pass, argv[4] = an existing pmc_traffic.json to merge into (configs are keyed by "<arch> <qtype>")
cfg = sys.argv[3] if len(sys.argv) > 3 else "large-v3 q5_0"
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --arch A --qtype Q --steps 1 --warmup 0 --n-decode 8`, one entry per "
                 "configuration; bytes = counter * 1024, FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md HBM section); "
                 "WRITE_SIZE uncalibrated", "round": rnd, "configs": {}}
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    old = json.load(open(sys.argv[4]))
    doc["configs"] = old.get("configs", {})
doc["configs"][cfg] = {"round": rnd, "kernels": kern}
print(json.dumps(doc, indent=1))
