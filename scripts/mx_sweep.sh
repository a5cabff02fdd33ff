#!/bin/bash
# sweep of the rendezvous window and the chain split with the matrix-core mat-vecs
cd "$(dirname "$0")/.." 2>/dev/null || cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05; mkdir -p $OUT
run() { echo "# $*"; env "$@" timeout 900 python3 scripts/stream_scaling.py --streams $S --batching 1 --steps 2 2>&1 | grep -v '^{"arch"' | cut -c1-220; }
{
  S=16,32 run GGML_MI355X_BATCH_WINDOW_US=1000
  S=16,32 run GGML_MI355X_BATCH_WINDOW_US=400
  S=16,32 run GGML_MI355X_BATCH_WINDOW_US=1000 GGML_MI355X_BATCH_COLS=32
  S=16,32 run GGML_MI355X_BATCH_WINDOW_US=400 GGML_MI355X_BATCH_COLS=32
  S=32    run GGML_MI355X_BATCH_WINDOW_US=1000 GGML_MI355X_BATCH_SPLIT_PCT=34
  S=16,32 run GGML_MI355X_BATCH_WINDOW_US=1000 GGML_MI355X_BATCH_SPLIT_PCT=50
} > $OUT/mx_sweep.txt 2>&1
cat $OUT/mx_sweep.txt
