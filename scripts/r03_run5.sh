#!/bin/bash
# Round 3, GPU call 5: LayerNorm inside the plane mat-vec + one-launch cross-attention — kernel tests, then A-B (new | LN separate | cross-attention
# as partial records | both old) on 5-token steps and on 8 / 16 batched streams, per-kernel profile of the new 5-token step
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
stage() { echo; echo "=== $1 === $(date +%T)"; }
STAGES=${*:-batchtests ab profile}
for s in $STAGES; do case $s in
batchtests)
    stage "tests/test_gpu_batch.py"
    timeout 900 python3 -m pytest tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 300 --timeout-method=thread > "$OUT/r03_pytest_batch.txt" 2>&1
    echo "exit=$?"; tail -12 "$OUT/r03_pytest_batch.txt"
    ;;
ab)
    stage "A-B: LN_FUSED x ATTN_PLANES_MAX_KV"
    for cfg in "1 1536" "0 1536" "1 512" "0 512"; do
        set -- $cfg
        export GGML_MI355X_LN_FUSED=$1 GGML_MI355X_ATTN_PLANES_MAX_KV=$2
        timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 --no-profile > "$OUT/r03_bench_ln$1_kv$2.json" 2> "$OUT/r03_bench_ln$1_kv$2.err"
        python3 -c "
import json; d=json.load(open('$OUT/r03_bench_ln$1_kv$2.json')); print('ln_fused=$1 attn_planes_max_kv=$2: ms/chunk', d['value'], 'decode ms/token', d['decode_ms_per_token'], 'batchd ms/token', d.get('batchd_ms_per_token'))"
        timeout 300 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 8,16 --batching 1 --n-decode 256 --steps 2 > "$OUT/r03_ab_streams_ln$1_kv$2.txt" 2>&1
        grep -v '"rows"' "$OUT/r03_ab_streams_ln$1_kv$2.txt" | cut -c1-120
    done
    unset GGML_MI355X_LN_FUSED GGML_MI355X_ATTN_PLANES_MAX_KV
    ;;
profile)
    stage "5-token steps, per kernel"
    timeout 300 python3 bench.py --profile-only --profile-what batchd > "$OUT/r03_profile_batchd_v3.json" 2> "$OUT/r03_profile_batchd_v3.err"
    python3 - <<'P'
import json
d=json.load(open("gpurun_out/r03_profile_batchd_v3.json"))
tot=sum(k["total_ms"] for k in d["kernels"])
print("5-token steps: total GPU ms over 16 steps:", round(tot,3), "-> per step", round(tot/16,4))
for k in d["kernels"][:12]: print(f'{k["name"][:70]:70s} {k["calls"]:6d} {k["total_ms"]*1e3/max(k["calls"],1):8.2f} us')
P
    ;;
pytest)
    stage "pytest -m gpu (everything)"
    timeout 1700 python3 -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --timeout-method=thread > "$OUT/r03_pytest_gpu.txt" 2>&1
    echo "exit=$?"; tail -30 "$OUT/r03_pytest_gpu.txt"
    ;;
esac; done
echo; echo "=== done $(date +%T)"
