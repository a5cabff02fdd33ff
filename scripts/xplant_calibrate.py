#!/usr/bin/env python3
"""TOOLING: calibration of the cross-attention-carried planted models (scripts/synth_model.py: XPLANT) with the REFERENCE CPU path only,
in the setting the test uses: whisper_full() greedy and beam-5, self-fed tokens (tests/native/bin/full_parity in its CPU-against-CPU
self-test mode).  For (f, g, c, w) builds the model and reports which of the two planted candidates every position chose and the
margin distribution of the greedy run.   python scripts/xplant_calibrate.py base.en q5_0 20,80,1,0.1 [max_tokens]
(Round 4's first calibration used teacher-forced random tokens and a random mel — model_parity — and did not carry over to
whisper_full: the encoder output of the real input moved w . enc_k, and with it the sign of the deciding number.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))
import synth_model as sm  # noqa: E402

arch, qtype, params = sys.argv[1], sys.argv[2], sys.argv[3]
max_tokens = sys.argv[4] if len(sys.argv) > 4 else "100"
os.environ["XPLANT_PARAMS"] = params
out = Path(f"/tmp/xplant_cal/{params.replace(',', '_')}")
m = sm.make_model(arch, qtype, out_dir=out, plant="x")
n_text_ctx = sm.ARCHS[arch][5]
te = sm.read_token_embedding(out / f"synth-{arch}-xplanted-f16.bin")
u, xa, xb = sm.xplant_tables(te, n_text_ctx, 1234)
xa, xb = list(xa), list(xb)
env = dict(os.environ, GGML_MI355X_PLUGIN="cpu", LD_LIBRARY_PATH=str(ROOT / "oracle" / "_ref"))
r = subprocess.run([str(ROOT / "tests/native/bin/full_parity"), str(m), max_tokens], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
assert r.returncode == 0, r.stderr[-1000:]
d = json.loads(r.stdout)
for mode in ("greedy", "beam5"):
    toks = d[mode]["cpu"]
    p0 = next((p for p in range(8) if toks and toks[0] in (xa[p], xb[p])), None)
    kind = "".join("a" if p0 is not None and t == xa[p0 + i] else ("b" if p0 is not None and t == xb[p0 + i] else "?") for i, t in enumerate(toks))
    print(f"{mode:7s} n {len(toks)} first position {p0}: a {kind.count('a')} b {kind.count('b')} other {kind.count('?')}   {kind}")
g = d["greedy"]
print("greedy:", {k: v for k, v in g.items() if k not in ("cpu", "gpu")})
