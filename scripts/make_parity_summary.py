#!/usr/bin/env python3
"""profiles/rNN_parity_summary.json from the per-model JSON files the GPU tests leave in gpurun_out/ (tests/native/model_parity.cpp,
full_parity.cpp, lang_detect.cpp).   usage: make_parity_summary.py <gpurun_out> > profiles/archive/r02_parity_summary.json"""
import json
import sys
from pathlib import Path

out = Path(sys.argv[1])
res = {"what": "model-level parity of every BASELINE configuration (tests/test_gpu.py::test_plugin_model_parity[_reference_exact_mode], "
               "tests/native/model_parity.cpp): the same model file through the unmodified libwhisper on the reference CPU backend and on the MI355X "
               "plugin; 128 teacher-forced single-token steps, a 5-token and a 48-token batch, a free-running greedy decode in lockstep. NMSE of the logits rows.",
       "models": {}, "whisper_full_pipeline": {}, "language_detection": {}}
for f in sorted(out.glob("model_parity_*.json")):
    d = json.loads(f.read_text())
    if "single" not in d:
        continue
    s, g = d["single"], d["greedy"]
    res["models"][f.stem[len("model_parity_"):]] = {
        "flash_attn": d["flash_attn"], "single_worst_nmse": s["worst_nmse"], "single_mean_nmse": s["mean_nmse"],
        "argmax_agree": f"{s['argmax_agree']}/{s['steps']}", "near_tie_steps": s["near_tie_steps"],
        "max_margin_over_maxdiff_on_mismatch": s["max_margin_over_maxdiff_on_mismatch"],
        "batch5_nmse": d["batch5"]["nmse"], "batch48_nmse": d["batch48"]["nmse"],
        "greedy_identical_prefix": f"{g['identical_prefix']}/{g['steps']}", "greedy_divergence_margin": g["divergence_margin"],
        "greedy_divergence_max_diff": g["divergence_max_diff"]}
for f in sorted(out.glob("full_parity_*.json")):
    d = json.loads(f.read_text())
    keys = ("n_cpu", "n_gpu", "identical_prefix", "steps_compared", "min_margin", "max_logit_diff", "divergence_margin", "divergence_logit_diff")
    res["whisper_full_pipeline"][f.stem[len("full_parity_"):]] = {m: {k: d[m][k] for k in keys if k in d[m]} for m in ("greedy", "beam5") if m in d}
for f in sorted(out.glob("lang_detect_*.json")):
    res["language_detection"][f.stem[len("lang_detect_"):]] = json.loads(f.read_text())
print(json.dumps(res, indent=1))
