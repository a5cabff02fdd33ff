#!/bin/bash
# Round 3, GPU call 3: where does an 8-column batched step spend its time (rocprofv3 kernel trace), logits mirror / planes-out geometry A-Bs
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
stage() { echo; echo "=== $1 === $(date +%T)"; }
STAGES=${*:-quick ab prof single}
for s in $STAGES; do case $s in
quick)
    stage "kernel tests + harness bit-identity (new build)"
    timeout 900 python3 -m pytest tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 300 --timeout-method=thread > "$OUT/r03_pytest_batch.txt" 2>&1
    echo "exit=$?"; tail -6 "$OUT/r03_pytest_batch.txt"
    timeout 600 python3 -m pytest tests/test_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 --timeout-method=thread -k "model_parity and (base or micro or turbo) or bench_smoke or native_harness" > "$OUT/r03_pytest_subset.txt" 2>&1
    echo "exit=$?"; tail -6 "$OUT/r03_pytest_subset.txt"
    ;;
ab)
    stage "8 streams batched: logits mirror on / off, planes-out rows 4 / 2"
    for cfg in "1 4" "0 4" "1 2"; do
        set -- $cfg
        GGML_MI355X_LOGITS_MIRROR=$1 GGML_MI355X_POUT_ROWS=$2 timeout 300 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 8 --batching 1 --n-decode 256 --steps 2 > "$OUT/r03_ab_mirror$1_pout$2.txt" 2>&1
        echo "mirror=$1 pout_rows=$2: $(grep -v '"rows"' "$OUT/r03_ab_mirror$1_pout$2.txt" | cut -c1-120)"
    done
    ;;
prof)
    stage "rocprofv3 --kernel-trace --stats: 8 streams, batched, 1 chunk each"
    rm -rf "$OUT/prof_batch8"
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof_batch8" -o b8 -- python3 "$ROOT/scripts/stream_scaling.py" --arch large-v3 --qtype q5_0 --streams 8 --batching 1 --n-decode 256 --steps 1 \
        > "$OUT/prof_batch8.json" 2> "$OUT/prof_batch8.err" )
    echo "exit=$?"
    f=$(find "$OUT/prof_batch8" -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" "$OUT/r03_kernel_stats_batch8_large-v3_q5_0.csv"; head -22 "$f" | cut -c1-170; fi
    python3 scripts/summarize_trace.py "$OUT/prof_batch8" > "$OUT/r03_kernel_trace_summary_batch8.txt" 2>&1; grep -v "at::\|rocclr" "$OUT/r03_kernel_trace_summary_batch8.txt" | head -30
    find "$OUT/prof_batch8" -name "*kernel_trace.csv" -size +20M -delete
    ;;
single)
    stage "single stream headline (mirror on / off)"
    for m in 1 0; do
        GGML_MI355X_LOGITS_MIRROR=$m timeout 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --multi-stream 0 --no-profile > "$OUT/r03_bench_mirror$m.json" 2> "$OUT/r03_bench_mirror$m.err"
        python3 -c "
import json; d=json.load(open('$OUT/r03_bench_mirror$m.json')); print('mirror=$m: ms/chunk', d['value'], 'encode', d['encode_ms'], 'decode ms/token', d['decode_ms_per_token'], 'batchd', d['batchd_ms_per_token'], 'get_tensor ms', d['backend']['host_ms_in_timed_region']['get_tensor'])"
    done
    ;;
esac; done
echo; echo "=== done $(date +%T)"
