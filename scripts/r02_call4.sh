#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
timeout 2700 python3 -m pytest tests -m gpu -q -p no:cacheprovider -rf > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit=$?"; tail -30 $OUT/pytest_gpu.txt | cut -c1-300
date +%T
SWEEP_STEPS=2 SWEEP_ARGS="--multi-stream 0" timeout 900 scripts/env_sweep.sh "GGML_MI355X_STEP_BLOCK=1" "GGML_MI355X_STEP_BLOCK=0" "GGML_MI355X_STEP_BLOCK=1 GGML_MI355X_GRAPH_SEG0=24" "GGML_MI355X_STEP_BLOCK=1 GGML_MI355X_GRAPH_SEGS=12,256" 2>&1 | tee $OUT/env_sweep_r02b.txt
date +%T
timeout 900 python3 scripts/stream_scaling.py large-v3 q5_0 1 2 4 6 8 12 2>&1 | tail -8 | tee $OUT/stream_scaling.txt
date +%T
