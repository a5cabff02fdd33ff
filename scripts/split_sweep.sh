#!/bin/bash
cd "$(dirname "$0")/.." 2>/dev/null || cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
{ for cfg in "67 8" "67 4" "50 4" "75 8" "40 4" "60 6"; do set -- $cfg
    echo "# GGML_MI355X_BATCH_SPLIT_PCT=$1 GGML_MI355X_BATCH_SPLIT_MIN=$2"
    GGML_MI355X_BATCH_SPLIT_PCT=$1 GGML_MI355X_BATCH_SPLIT_MIN=$2 timeout 600 python3 scripts/stream_scaling.py --streams 8,16,32 --batching 1 --steps 2 2>&1 | grep -v '^{"arch"' | cut -c1-200
  done; } > $OUT/split_sweep.txt 2>&1
cat $OUT/split_sweep.txt
