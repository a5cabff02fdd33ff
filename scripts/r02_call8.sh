#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
for cfg in "1 8" "0 8" "1 4" "0 4"; do set -- $cfg
  echo "GRAPHS=$1 HW_QUEUES=$2"
  GGML_MI355X_GRAPHS=$1 STREAM_HW_QUEUES=$2 GPU_MAX_HW_QUEUES=$2 timeout 600 python3 scripts/stream_scaling.py large-v3 q5_0 1 2 4 8 2>&1 | tail -1 | cut -c1-700
done | tee $OUT/stream_scaling_graphs_vs_eager.txt
date +%T
