#!/usr/bin/env python3
"""Concurrent streams on ONE MI355X through the native harness (include/mi355x_host.h): chunks/s per stream count, with the plugin's
cross-state batching off and / or on.
   usage: scripts/stream_scaling.py [--arch large-v3] [--qtype q5_0] [--streams 1,2,4,8] [--batching 0,1] [--n-decode 256] [--steps 2]
   env:   GPU_MAX_HW_QUEUES (8), GGML_MI355X_BATCH_COLS (fixed chain width), GGML_MI355X_MX_MIN_T (0: mat-vecs of wide chains off the matrix cores)"""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "scripts"))
os.environ.setdefault("GGML_MI355X_STRICT", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("STREAM_HW_QUEUES", "8"))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from synth_model import make_model  # noqa: E402
from whisper_cpp_amd import host_api as h  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="large-v3")
ap.add_argument("--qtype", default="q5_0")
ap.add_argument("--streams", default="1,2,4,6,8,12")
ap.add_argument("--batching", default="0")
ap.add_argument("--n-decode", type=int, default=256)
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
m = make_model(a.arch, a.qtype)
rows = []
for batching in [int(x) for x in a.batching.split(",")]:
    for s in [int(x) for x in a.streams.split(",")]:
        r = h.run(m, use_gpu=True, n_devices=1, streams=s, n_decode=a.n_decode, steps=a.steps, warmup=1, batching=batching)
        rows.append({"batching": batching, "streams": s, "chunks_per_s": round(r["chunks_per_s"], 3), "ms_per_chunk_per_stream": round(r["ms_per_chunk_per_stream"], 1),
                     "batch_stats": r["batch_stats"], "rc": r["rc"], "error": r["error"]})
        print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"arch": a.arch, "qtype": a.qtype, "n_decode": a.n_decode, "hw_queues": os.environ["GPU_MAX_HW_QUEUES"],
                  "batch_cols": os.environ.get("GGML_MI355X_BATCH_COLS", "default"), "mx_min_t": os.environ.get("GGML_MI355X_MX_MIN_T", "default"), "rows": rows}))
