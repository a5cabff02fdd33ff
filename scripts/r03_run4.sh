#!/bin/bash
# Round 3, GPU call 4: full GPU suite on the new build; A-Bs: self-attention straight to planes, logits mirror, vocabulary-projection passes
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
stage() { echo; echo "=== $1 === $(date +%T)"; }
STAGES=${*:-batchtests ab single pytest}
for s in $STAGES; do case $s in
batchtests)
    stage "tests/test_gpu_batch.py"
    timeout 900 python3 -m pytest tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 300 --timeout-method=thread > "$OUT/r03_pytest_batch.txt" 2>&1
    echo "exit=$?"; tail -12 "$OUT/r03_pytest_batch.txt"
    ;;
ab)
    stage "8 streams batched: self-attention planes x logits mirror"
    for cfg in "1 1" "0 1" "1 0"; do
        set -- $cfg
        GGML_MI355X_SELF_ATTN_PLANES=$1 GGML_MI355X_LOGITS_MIRROR=$2 timeout 300 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 8 --batching 1 --n-decode 256 --steps 2 > "$OUT/r03_ab_selfq$1_mirror$2.txt" 2>&1
        echo "self_attn_planes=$1 mirror=$2: $(grep -v '"rows"' "$OUT/r03_ab_selfq$1_mirror$2.txt" | cut -c1-110)"
    done
    timeout 300 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 16 --batching 1 --n-decode 256 --steps 2 > "$OUT/r03_ab_16streams.txt" 2>&1
    echo "16 streams (8 columns x 2 chains): $(grep -v '"rows"' "$OUT/r03_ab_16streams.txt" | cut -c1-110)"
    ;;
single)
    stage "single stream: logits mirror, vocabulary-projection passes, 5-token steps"
    for cfg in "1 32" "0 32" "1 16" "1 8"; do
        set -- $cfg
        GGML_MI355X_LOGITS_MIRROR=$1 GGML_MI355X_GEMV_PASS_WAVES=$2 timeout 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --multi-stream 0 --no-profile > "$OUT/r03_bench_mirror$1_pw$2.json" 2> "$OUT/r03_bench_mirror$1_pw$2.err"
        python3 -c "
import json; d=json.load(open('$OUT/r03_bench_mirror$1_pw$2.json')); print('mirror=$1 pass_waves=$2: ms/chunk', d['value'], 'encode', d['encode_ms'], 'decode ms/token', d['decode_ms_per_token'], 'batchd', d['batchd_ms_per_token'], 'prompt', d['prompt_ms_per_token'], 'get_tensor ms', d['backend']['host_ms_in_timed_region']['get_tensor'])"
    done
    timeout 300 python3 bench.py --profile-only --profile-what batchd > "$OUT/r03_profile_batchd_v2.json" 2> "$OUT/r03_profile_batchd_v2.err"
    python3 - <<'P'
import json
d=json.load(open("gpurun_out/r03_profile_batchd_v2.json"))
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
