#!/usr/bin/env python3
"""TOOLING: calibration of the cross-attention-carried planted models (scripts/synth_model.py: XPLANT) with the REFERENCE CPU path only
(tests/native/bin/model_parity in its CPU-against-CPU self-test mode): for (f, g, c) builds the model, runs N teacher-forced steps on
two different inputs ("audios") and reports how many steps choose one of the two planted candidates, the margin distribution and how
many decisions differ between the two inputs.   python scripts/xplant_calibrate.py base.en q5_0 6,24,3 [steps]"""
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
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 128
os.environ["XPLANT_PARAMS"] = params
out = Path(f"/tmp/xplant_cal/{params.replace(',', '_')}")
m = sm.make_model(arch, qtype, out_dir=out, plant="x")
(n_vocab, _, _, _, _, n_text_ctx, n_ts, _, n_tl, _) = sm.ARCHS[arch]
te = sm.read_token_embedding(out / f"synth-{arch}-xplanted-f16.bin")
u, xa, xb = sm.xplant_tables(te, n_text_ctx, 1234)
env = dict(os.environ, GGML_MI355X_PLUGIN="cpu", MODEL_PARITY_ALL_STEPS="1", LD_LIBRARY_PATH=str(ROOT / "oracle" / "_ref"))
runs = []
for seed in (42, 7):
    r = subprocess.run([str(ROOT / "tests/native/bin/model_parity"), str(m), str(steps)], env=dict(env, MODEL_PARITY_MEL_SEED=str(seed)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    runs.append(json.loads(r.stdout))
res = []
for d in runs:
    toks = [s["tok_cpu"] for s in d["steps"]]
    marg = np.array([s["margin"] for s in d["steps"]])
    kind = ["a" if t == xa[p] else ("b" if t == xb[p] else "?") for p, t in enumerate(toks)]
    res.append((toks, marg, kind))
    print(f"steps {len(toks)}: a {kind.count('a')}  b {kind.count('b')}  other {kind.count('?')};  margin min {marg.min():.3f}  p10 {np.percentile(marg, 10):.3f}  median {np.median(marg):.3f};  max |logit| {d['single']['max_abs_logit']:.1f}")
diff = sum(1 for x, y in zip(res[0][2], res[1][2]) if x != y)
print(f"decisions that differ between the two inputs: {diff} of {len(res[0][2])}")
print("pattern A:", "".join(res[0][2][:96]))
print("pattern B:", "".join(res[1][2][:96]))
