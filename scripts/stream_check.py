#!/usr/bin/env python3
"""Concurrency check for the backend (SURVEY.md §8b "Threading"): S whisper_states on one whisper_context run the
whisper-bench protocol from S host threads at once; every stream's logits — EVERY decode step's row — must equal, bit for bit, what
the same stream produces when it runs alone.  From 5 streams on the concurrent leg runs as merged launch chains (cross-state batching,
on by default): the batched-versus-own-chain comparison of BASELINE.json configs[3].  Prints one JSON line.   usage: stream_check.py <arch> <qtype> <streams> <n_decode>"""
import ctypes as C
import json
import os
import sys
import threading
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GGML_MI355X_STRICT", "1")
import numpy as np  # noqa: E402

import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

graft.load_package()
from whisper_cpp_amd.streams import Streams  # noqa: E402
sys.path.insert(0, str(ROOT / "scripts"))
from synth_model import make_model  # noqa: E402

arch, qtype, S, n_dec = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
model = make_model(arch, qtype)
w, p = bench.load_host(ROOT / "whisper.cpp_amd" / "lib" / "libggml-mi355x.so")
cp = w.whisper_context_default_params()
cp.use_gpu, cp.flash_attn, cp.gpu_device = True, True, 0
ctx = w.whisper_init_from_file_with_params(str(model).encode(), cp)
assert ctx, "model load failed"
n_mels = w.whisper_model_n_mels(ctx)
mels = [(np.random.default_rng(7 + i).random((n_mels, 3000), dtype=np.float32) * 2 - 1) for i in range(S)]
st = Streams(w, ctx, S, mels)
rng = np.random.default_rng(3)
toks = [(C.c_int32 * 512)(*[int(t) for t in rng.integers(0, 50000, 512)]) for _ in range(S)]

serial = [[] for _ in range(S)]
for s in range(S):
    st.chunk_one(s, n_dec, tokens=toks[s], keep_logits=serial[s])
conc = [[] for _ in range(S)]
errs = []


def run(s):
    try:
        for _ in range(2):                    # second round replays cached hipGraphs on every stream at once
            conc[s].clear()
            st.chunk_one(s, n_dec, tokens=toks[s], keep_logits=conc[s])
    except Exception as e:  # noqa: BLE001
        errs.append(repr(e))


th = [threading.Thread(target=run, args=(s,)) for s in range(S)]
for t in th:
    t.start()
for t in th:
    t.join()
mismatch = sum(int(not np.array_equal(a.view(np.uint32), b.view(np.uint32))) for s in range(S) for a, b in zip(serial[s], conc[s]))
distinct = int(not np.array_equal(serial[0][-1], serial[-1][-1])) if S > 1 else 1
finite = bool(all(np.isfinite(x).all() for s in range(S) for x in conc[s]))
# merged launch chains the plugin formed for the concurrent leg (cross-state batching is on by default from 5 decoding states)
bs = (C.c_uint64 * 5)()
p.ggml_backend_mi355x_batch_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
p.ggml_backend_mi355x_batch_stats(0, bs)
print(json.dumps({"streams": S, "n_decode": n_dec, "rows_compared": sum(len(x) for x in conc), "mismatching_rows": mismatch,
                  "streams_differ_from_each_other": distinct, "finite": finite, "errors": errs,
                  "batch_stats": {"chains": int(bs[0]), "columns": int(bs[1]), "solo": int(bs[2]), "fallbacks": int(bs[3]), "timeouts": int(bs[4])}}))
st.close()
w.whisper_free(ctx)
