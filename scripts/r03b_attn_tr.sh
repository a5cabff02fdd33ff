#!/bin/bash
# GPU check of the transposing-read V path of the attention kernel: tests, kernel timings, whole encode A-B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu.py -x -q -m gpu --timeout 600 -k "key_groups or flash_attn" 2>&1 | tail -5
python3 scripts/enc_kbench.py --what attn --iters 40 2>&1 | grep -v amdgpu.ids
python3 scripts/enc_ab.py --reps 8 -- "" "GGML_MI355X_FATTN_TR=0" "" "GGML_MI355X_FATTN_TR=0" 2>&1 | grep -v "^whisper_\|^ggml_\|load_backend"
