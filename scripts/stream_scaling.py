#!/usr/bin/env python3
"""Concurrent streams on ONE MI355X through the native harness (include/mi355x_host.h): chunks/s for 1, 2, 4, 6, 8, 12 streams.
   usage: scripts/stream_scaling.py [arch=large-v3] [qtype=q5_0] [streams...]"""
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

arch = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
qtype = sys.argv[2] if len(sys.argv) > 2 else "q5_0"
counts = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 6, 8, 12]
m = make_model(arch, qtype)
rows = []
for s in counts:
    r = h.run(m, use_gpu=True, n_devices=1, streams=s, n_decode=256, steps=2, warmup=1)
    rows.append({"streams": s, "chunks_per_s": round(r["chunks_per_s"], 3), "ms_per_chunk_per_stream": round(r["ms_per_chunk_per_stream"], 1), "rc": r["rc"], "error": r["error"]})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"arch": arch, "qtype": qtype, "hw_queues": os.environ["GPU_MAX_HW_QUEUES"], "rows": rows}))
