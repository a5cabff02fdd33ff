#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ echo "deferred uploads (default), GPU_MAX_HW_QUEUES=8"; STREAM_HW_QUEUES=8 timeout 600 python3 scripts/stream_scaling.py large-v3 q5_0 1 2 3 4 5 6 8 2>/dev/null | tail -1
  echo "deferred uploads (default), GPU_MAX_HW_QUEUES=4"; STREAM_HW_QUEUES=4 timeout 600 python3 scripts/stream_scaling.py large-v3 q5_0 4 6 8 12 2>/dev/null | tail -1
  echo "GGML_MI355X_DEFER_IO=0, GPU_MAX_HW_QUEUES=8"; GGML_MI355X_DEFER_IO=0 STREAM_HW_QUEUES=8 timeout 600 python3 scripts/stream_scaling.py large-v3 q5_0 3 4 5 2>/dev/null | tail -1
} | tee $OUT/stream_scaling_defer_io.txt
