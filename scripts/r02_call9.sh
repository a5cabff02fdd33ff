#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
GGML_MI355X_QATTN_WAVES=16 timeout 600 python3 -m pytest tests -m gpu -q -p no:cacheprovider -k "fused_ln_q_attention" 2>&1 | tail -3
SWEEP_STEPS=3 SWEEP_ARGS="--multi-stream 0" timeout 900 scripts/env_sweep.sh "X=0" "GGML_MI355X_QATTN_WAVES=16" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "GGML_MI355X_QATTN=0" "X=1" 2>&1 | tee $OUT/env_sweep_r02d.txt
date +%T
bash scripts/gpu_round.sh pytest bench prof pmc wbench > $OUT/round_full.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.txt | tail -8
cut -c1-1200 $OUT/bench_large-v3_q5_0.json
tail -12 $OUT/round_full.log | cut -c1-200
date +%T
