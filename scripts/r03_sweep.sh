#!/bin/bash
# Round-3 (second session) GPU call 1: the new encoder kernels — parity tests, kernel-level timings of every variant, whole-encode A-B in one
# process, and the cache-level experiment of the decode mat-vec (what an L2 / Infinity-Cache-resident weight matrix would buy).
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
stage() { echo; echo "=== $1 === $(date +%T)"; }
stage "pytest: new encoder kernels + the existing attention / GEMM tests"
timeout 900 python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu --timeout 600 2>&1 | tail -15 | tee "$OUT/r03b_pytest_encoder.txt"
timeout 600 python3 -m pytest tests/test_gpu.py -q -m gpu --timeout 600 -k "flash_attn or ring or gemm or layernorm or mul_mat_vs_oracle" 2>&1 | tail -6 | tee "$OUT/r03b_pytest_attn_gemm.txt"
stage "encoder kernels, every variant (scripts/enc_kbench.py)"
timeout 600 python3 scripts/enc_kbench.py --iters 40 2>&1 | tee "$OUT/r03b_encoder_kernel_variants.txt"
stage "whole encode, one process (scripts/enc_ab.py)"
timeout 900 python3 scripts/enc_ab.py --reps 8 -- "" "GGML_MI355X_FATTN_NG=1" "GGML_MI355X_FATTN_NG=2" "GGML_MI355X_FATTN_NG=3" "GGML_MI355X_FATTN_NG=4" \
    "GGML_MI355X_GEMM_RING_TM256=1282" "GGML_MI355X_GEMM_RING_TM256=1283" "GGML_MI355X_GEMM_RING_TM256=642 GGML_MI355X_GEMM_RING_TM256_MIN=100" \
    "GGML_MI355X_GEMM_GROUP_CFG=2561282" "GGML_MI355X_GEMM_GROUP_CFG=2561283" "GGML_MI355X_GEMM_GROUP_CFG=256642" "GGML_MI355X_GEMM_GROUP_CFG=256643" \
    "GGML_MI355X_FATTN_NG=3 GGML_MI355X_GEMM_RING_TM256=1282 GGML_MI355X_GEMM_GROUP_CFG=2561282" \
    "GGML_MI355X_FATTN_NG=3 GGML_MI355X_GEMM_RING_TM256=1283 GGML_MI355X_GEMM_GROUP_CFG=256642" "" 2>&1 | grep -v "^whisper_\|^ggml_" | tee "$OUT/r03b_encode_ab.txt"
timeout 300 python3 scripts/enc_ab.py --arch large-v3-turbo --qtype q8_0 --reps 8 -- "" "GGML_MI355X_FATTN_NG=1" "GGML_MI355X_FATTN_NG=3" 2>&1 | grep -v "^whisper_\|^ggml_" | tee -a "$OUT/r03b_encode_ab.txt"
stage "decode mat-vec by residency of its weights (scripts/kbench.py --cycle)"
timeout 300 python3 scripts/kbench.py --cycle --iters 200 2> "$OUT/r03b_kbench_cycle.err" | python3 -c "
import json, sys
d = json.load(sys.stdin)
for c in d['cases']:
    print(f\"{c['case']:44s} {c['us_per_call_events']:8.2f} us  {c['algo_MB']:8.3f} MB  {c['GBps']} GB/s\")" | tee "$OUT/r03b_kbench_cycle.txt"
echo; echo "=== done $(date +%T)"
