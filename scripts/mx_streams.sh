#!/bin/bash
# A-B of the matrix-core mat-vec (decode_mx.hip) on whole merged chains: chunks/s at 8 .. 32 streams, chain widths default / fixed
cd "$(dirname "$0")/.." 2>/dev/null || cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05; mkdir -p $OUT
run() { echo "# $*"; env "$@" timeout 900 python3 scripts/stream_scaling.py --streams $S --batching 1 --steps 2 2>&1 | grep -v '^{"arch"' | cut -c1-220; }
{
  S=8,16,24,32 run GGML_MI355X_MX_MIN_T=9
  S=16,32      run GGML_MI355X_MX_MIN_T=9 GGML_MI355X_BATCH_COLS=32
  S=32         run GGML_MI355X_MX_MIN_T=9 GGML_MI355X_BATCH_COLS=16
  S=16,32      run GGML_MI355X_MX_MIN_T=0
} > $OUT/mx_streams.txt 2>&1
cat $OUT/mx_streams.txt
