#!/bin/bash
# Round 3, GPU call 2: cross-state batching after the congruence fix (columns per chain x concurrent chains), next-stage weight prefetch A-B,
# the whole GPU test suite.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
stage() { echo; echo "=== $1 === $(date +%T)"; }
STAGES=${*:-probe harness scaling prefetch pytest}
for s in $STAGES; do case $s in
probe)
    stage "CU-mask probe"
    timeout 60 scripts/_bin/cumask_probe > "$OUT/r03_cumask_probe.txt" 2>&1; echo "exit=$?"; cat "$OUT/r03_cumask_probe.txt"
    ;;
harness)
    stage "cross-state batching: bit-identity through the harness"
    timeout 600 python3 -m pytest tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 300 --timeout-method=thread -k "cross_state_batching" > "$OUT/r03_pytest_batch_harness.txt" 2>&1
    echo "exit=$?"; tail -15 "$OUT/r03_pytest_batch_harness.txt"
    ;;
scaling)
    stage "columns per chain x concurrent chains"
    for cfg in "8 1,2,4,8,16" "4 4,8,16" "2 4,8"; do
        set -- $cfg
        GGML_MI355X_BATCH_COLS=$1 timeout 600 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams $2 --batching 1 --n-decode 256 --steps 2 > "$OUT/r03_stream_scaling_batch_cols$1.txt" 2>&1
        echo "cols $1 exit=$?"; grep -v '"rows"' "$OUT/r03_stream_scaling_batch_cols$1.txt"
    done
    GGML_MI355X_BATCH_COLS=8 timeout 300 python3 scripts/stream_scaling.py --arch large-v3-turbo --qtype q8_0 --streams 1,8 --batching 0,1 --n-decode 256 --steps 2 > "$OUT/r03_stream_scaling_batch_turbo.txt" 2>&1
    grep -v '"rows"' "$OUT/r03_stream_scaling_batch_turbo.txt"
    ;;
prefetch)
    stage "next-stage weight prefetch A-B (single stream, whisper-bench protocol)"
    for rep in 1 2; do for pf in 0 1; do
        GGML_MI355X_PREFETCH=$pf timeout 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --multi-stream 0 --no-profile > "$OUT/r03_bench_prefetch${pf}_$rep.json" 2> "$OUT/r03_bench_prefetch${pf}_$rep.err"
        python3 -c "
import json; d=json.load(open('$OUT/r03_bench_prefetch${pf}_$rep.json')); print('prefetch=$pf rep $rep: ms/chunk', d['value'], 'encode', d['encode_ms'], 'decode ms/token', d['decode_ms_per_token'], 'batchd', d['batchd_ms_per_token'], 'gpu_span ms', d['backend']['host_ms_in_timed_region']['gpu_span'])"
    done; done
    ;;
pytest)
    stage "pytest -m gpu (everything)"
    timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --timeout-method=thread > "$OUT/r03_pytest_gpu.txt" 2>&1
    echo "exit=$?"; tail -25 "$OUT/r03_pytest_gpu.txt"
    ;;
esac; done
echo; echo "=== done $(date +%T)"
