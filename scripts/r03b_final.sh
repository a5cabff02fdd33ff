#!/bin/bash
# Round-3 closing GPU call of the second session: full GPU suite, headline bench (roofline, cpu_baseline, multi_stream), rocprofv3 kernel trace + stats,
# PMC passes, stock whisper-bench + plugin, the other BASELINE configs, reduced stream-scaling tables.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
STAGES=${*:-main configs scaling}
stage() { echo; echo "=== $1 === $(date +%T)"; }
for s in $STAGES; do case $s in
main)
    PYTEST_ARGS="--timeout 900 --timeout-method=thread" bash scripts/gpu_round.sh pytest bench prof ${MAIN_STAGES:-pmc} wbench > "$OUT/round_full.log" 2>&1
    grep -E "passed|failed|^FAILED|^ERROR" "$OUT/pytest_gpu.txt" | tail -8
    cut -c1-900 "$OUT/bench_large-v3_q5_0.json"
    grep -E "time =" "$OUT/wbench_gpu_large-v3_q5_0.log"
    ;;
configs)
    stage "other configurations of BASELINE.json (3 steps each, no CPU leg)"
    for spec in "large-v3-turbo q8_0" "large-v3 q4_k" "large-v3 q8_0" "base.en q5_0" "tiny.en q5_0" "tiny.en f16"; do
        set -- $spec
        timeout 400 python3 bench.py --arch "$1" --qtype "$2" --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 > "$OUT/bench_$1_$2.json" 2> "$OUT/bench_$1_$2.err"
        python3 -c "
import json
try:
    d=json.load(open('$OUT/bench_$1_$2.json')); r=d.get('roofline') or {}
    print('$1 $2: ms/chunk', d['value'], 'encode', d['encode_ms'], 'decode ms/token', d['decode_ms_per_token'], 'batchd', d['batchd_ms_per_token'], 'prompt', d['prompt_ms_per_token'], '| roofline', r.get('kernel'), r.get('frac'))
except Exception as e: print('$1 $2: failed', e)"
    done
    ;;
scaling)
    stage "concurrent streams on one GPU: own chains | merged chains"
    timeout 600 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 1,4 --batching 0 --n-decode 256 --steps 2 > "$OUT/r03b_stream_scaling_unbatched.txt" 2>&1
    grep -v '"rows"' "$OUT/r03b_stream_scaling_unbatched.txt" | cut -c1-130
    timeout 600 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 8,12 --batching 1 --n-decode 256 --steps 2 > "$OUT/r03b_stream_scaling_batched.txt" 2>&1
    grep -v '"rows"' "$OUT/r03b_stream_scaling_batched.txt" | cut -c1-130
    timeout 400 python3 scripts/stream_scaling.py --arch large-v3-turbo --qtype q8_0 --streams 16 --batching 1 --n-decode 256 --steps 2 > "$OUT/r03b_stream_scaling_turbo.txt" 2>&1
    grep -v '"rows"' "$OUT/r03b_stream_scaling_turbo.txt" | cut -c1-130
    ;;
esac; done
echo; echo "=== done $(date +%T)"
