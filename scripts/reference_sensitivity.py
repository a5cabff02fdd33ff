#!/usr/bin/env python3
"""How far does the REFERENCE move against itself?  (no plugin, no GPU: tests/native/bin/model_parity and layer_bisect self-tests)
  * mel input scaled by (1 + eps), eps = 1e-7 and 1e-6: one f32 rounding and ten
  * 8 threads against 2 threads (its single-token flash attention is split over the threads, ggml-cpu/ops.cpp:9117-9150)
Writes profiles/archive/r02_reference_self_sensitivity.json — the floor the model-level parity tolerances are set against."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))
from synth_model import make_model  # noqa: E402

env0 = dict(os.environ, GGML_MI355X_PLUGIN="cpu", LD_LIBRARY_PATH=str(ROOT / "oracle" / "_ref"))
out = {"what": __doc__, "cases": []}
for arch, qtype in (("base.en", "q5_0"), ("base.en", "q8_0"), ("base.en", "f16")):
    m = make_model(arch, qtype)
    for eps in ("1e-7", "1e-6"):
        r = subprocess.run([str(ROOT / "tests/native/bin/model_parity"), str(m), "64"], env=dict(env0, MODEL_PARITY_PERTURB=eps), stdout=subprocess.PIPE, text=True, check=True)
        d = json.loads(r.stdout)
        out["cases"].append({"model": f"{arch} {qtype}", "perturbation": f"mel * (1 + {eps})", "single_token_logits_nmse_mean": d["single"]["mean_nmse"],
                             "single_token_logits_nmse_worst": d["single"]["worst_nmse"], "argmax_agree": f"{d['single']['argmax_agree']}/{d['single']['steps']}",
                             "batch5_nmse": d["batch5"]["nmse"], "batch48_nmse": d["batch48"]["nmse"],
                             "free_running_greedy_identical_prefix": f"{d['greedy']['identical_prefix']}/{d['greedy']['steps']}"})
    r = subprocess.run([str(ROOT / "tests/native/bin/layer_bisect"), str(m), "1", "3"], env=dict(env0, BISECT_THREADS="8", BISECT_THREADS_B="2"), stdout=subprocess.PIPE, text=True, check=True)
    d = json.loads(r.stdout)
    out["cases"].append({"model": f"{arch} {qtype}", "perturbation": "8 threads vs 2 threads (single-token step, n_past = 3)", "logits_nmse": d["logits_nmse"],
                         "flash_attn_nodes_worst_nmse": d["per_op"]["FLASH_ATTN_EXT"]["worst_nmse"]})
# whisper_full() end to end (mel front end, greedy and 5-beam search): the reference against itself on the PCM scaled by (1 + eps)
for arch, qtype in (("micro", "q5_0"), ("base.en", "q5_0"), ("base.en", "q8_0")):
    m = make_model(arch, qtype)
    for eps in ("1e-6", "1e-4"):
        r = subprocess.run([str(ROOT / "tests/native/bin/full_parity"), str(m), "48"], env=dict(env0, FULL_PARITY_PERTURB=eps), stdout=subprocess.PIPE, text=True, check=True)
        d = json.loads(r.stdout)
        out["cases"].append({"model": f"{arch} {qtype}", "perturbation": f"whisper_full on pcm * (1 + {eps})",
                             "greedy_identical_prefix": f"{d['greedy']['identical_prefix']}/{d['greedy']['n_cpu']}",
                             "beam5_identical_prefix": f"{d['beam5']['identical_prefix']}/{d['beam5']['n_cpu']}"})
(ROOT / "profiles" / "archive" / "r02_reference_self_sensitivity.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out["cases"], indent=1))
