#!/bin/bash
# One gpurun call around the > 8-column merged chains (images of 8 columns, decode_q.hip) and the x-planted token parity test.
# usage: scripts/gpu_cols.sh [stage ...]   stages: batchtest scaling xplant streams
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
STAGES=${*:-batchtest scaling xplant streams}
stage() { echo; echo "=== $1 === $(date +%T)"; }
for s in $STAGES; do case $s in
batchtest)
    stage "pytest tests/test_gpu_batch.py"
    timeout 1500 python3 -m pytest tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_batch.txt" 2>&1
    echo "exit=$?"; tail -30 "$OUT/pytest_batch.txt"
    ;;
scaling)
    stage "stream scaling, batched, large-v3 Q5_0"
    { echo "# GGML_MI355X_BATCH_COLS=16"; GGML_MI355X_BATCH_COLS=16 timeout 600 python3 scripts/stream_scaling.py --streams ${SCALING_16:-8,12,16} --batching 1 --steps 2 2>&1 | grep -v '^{"arch"'
      echo "# GGML_MI355X_BATCH_COLS=32"; GGML_MI355X_BATCH_COLS=32 timeout 600 python3 scripts/stream_scaling.py --streams ${SCALING_32:-16,24,32} --batching 1 --steps 2 2>&1 | grep -v '^{"arch"'
    } > "$OUT/stream_scaling_cols.txt" 2>&1
    cut -c1-260 "$OUT/stream_scaling_cols.txt"
    ;;
xplant)
    stage "x-planted token parity + fault controls"
    timeout 2400 python3 -m pytest tests/test_gpu.py -m gpu -q -p no:cacheprovider -k "${XPLANT_SEL:-cross_attention_carried}" > "$OUT/pytest_xplant.txt" 2>&1
    echo "exit=$?"; tail -30 "$OUT/pytest_xplant.txt"
    ;;
streams)
    stage "concurrent streams == serial (incl. 12 and 16 streams)"
    timeout 1800 python3 -m pytest tests/test_gpu.py -m gpu -q -p no:cacheprovider -k "concurrent_streams" > "$OUT/pytest_streams.txt" 2>&1
    echo "exit=$?"; tail -30 "$OUT/pytest_streams.txt"
    ;;
scalingauto)
    stage "stream scaling with the default chain widths (+ cross-attention straight to planes at 16 streams)"
    { echo "# default"; timeout 900 python3 scripts/stream_scaling.py --streams ${AUTO_STREAMS:-8,12,16,24,32} --batching 1 --steps 2 2>&1 | grep -v '^{"arch"'
      echo "# GGML_MI355X_ATTN_PLANES_MAX_KV=1536"; GGML_MI355X_ATTN_PLANES_MAX_KV=1536 timeout 300 python3 scripts/stream_scaling.py --streams 16 --batching 1 --steps 2 2>&1 | grep -v '^{"arch"'
    } > "$OUT/stream_scaling_auto.txt" 2>&1
    cut -c1-260 "$OUT/stream_scaling_auto.txt"
    ;;
scaling16)
    stage "16 streams, columns per chain 8 / 10 / 12 / 16"
    { for cols in 8 10 12 16; do echo "# GGML_MI355X_BATCH_COLS=$cols"; GGML_MI355X_BATCH_COLS=$cols timeout 300 python3 scripts/stream_scaling.py --streams 16 --batching 1 --steps 2 2>&1 | grep -v '^{"arch"'; done; } > "$OUT/stream_scaling_16.txt" 2>&1
    cut -c1-260 "$OUT/stream_scaling_16.txt"
    ;;
trace)
    stage "rocprofv3 kernel trace of 16 batched streams (large-v3 Q5_0, 64 steps per chunk)"
    rm -rf "$OUT/trace16"; mkdir -p "$OUT/trace16"
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace16" -o t16 --output-format csv -- python3 "$ROOT/scripts/stream_scaling.py" --streams ${TRACE_STREAMS:-16} --batching 1 --steps 1 --n-decode 64 > "$OUT/trace16/run.log" 2>&1 )
    echo "exit=$?"; tail -3 "$OUT/trace16/run.log" | cut -c1-300
    f=$(find "$OUT/trace16" -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && { cp "$f" "$OUT/trace16_kernel_stats.csv"; head -25 "$f" | cut -c1-220; }
    find "$OUT/trace16" -name '*kernel_trace.csv' -size +60M -delete
    ;;
pmc16)
    stage "rocprofv3 --pmc FETCH_SIZE of 16 batched streams in ONE chain of 16 columns (are the weights read once per step?)"
    rm -rf "$OUT/pmc16"; mkdir -p "$OUT/pmc16"
    ( cd /tmp && GGML_MI355X_BATCH_COLS=16 GGML_MI355X_BATCH_WINDOW_US=500000 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc16" -o p16 --output-format csv -- python3 "$ROOT/scripts/stream_scaling.py" --streams 16 --batching 1 --steps 1 --n-decode 12 > "$OUT/pmc16/run.log" 2>&1 )
    echo "exit=$?"
    python3 scripts/summarize_pmc.py "$OUT/pmc16" > "$OUT/pmc16_FETCH_SIZE.summary.txt" 2>&1; grep -E "k_gemv_q|k_vocab|k_fattn_dec_multi|k_act_prepare|Kernel|kernel" "$OUT/pmc16_FETCH_SIZE.summary.txt" | head -20 | cut -c1-220
    find "$OUT/pmc16" -name "*.csv" -size +20M -delete
    ;;
xcd)
    stage "XCD-local teams probe"
    { timeout 120 scripts/probes/_bin/xcd_team_probe 32 11; echo; timeout 120 scripts/probes/_bin/xcd_team_probe 32 16; } 2>&1 | tee "$OUT/xcd_team_probe.txt"
    ;;
esac; done
echo; echo "=== done === $(date +%T)"
