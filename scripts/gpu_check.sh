#!/bin/bash
# One gpurun call = one full evidence sweep.  Everything lands in gpurun_out/ (merged back by gpurun).
# usage: scripts/gpu_check.sh [stage ...]   stages: info ops model bench prof pytest  (default: info ops model bench)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export GGML_BACKEND_PATH=$GGML_MI355X_PLUGIN
export LD_LIBRARY_PATH=$ROOT/oracle/_ref:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
REF=$ROOT/oracle/_ref
STAGES=${*:-info ops model bench}
NCPU=$(nproc)

stage() { echo; echo "=== $1 === $(date +%T)"; }

for s in $STAGES; do case $s in
info)
    stage info
    { rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -12; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; free -g | head -2; } > "$OUT/info.txt" 2>&1
    cat "$OUT/info.txt"
    ;;
ops)
    stage "op parity (node + sched modes)"
    GGML_MI355X_DEBUG=1 timeout 900 tests/native/bin/op_parity > "$OUT/op_parity.jsonl" 2> "$OUT/op_parity.err"
    echo "exit=$? lines=$(wc -l < "$OUT/op_parity.jsonl")"
    tail -5 "$OUT/op_parity.err"
    python3 scripts/summarize_ops.py "$OUT/op_parity.jsonl" | tee "$OUT/op_parity_summary.txt" | tail -40
    ;;
model)
    stage "model parity"
    for spec in "micro q5_0" "tiny.en f16" "base.en q5_0" "base.en q4_k" "base.en q8_0" "base.en q4_0" "large-v3-2l q5_0"; do
        set -- $spec
        m=$(python3 scripts/synth_model.py --arch "$1" --qtype "$2") || continue
        for fuse in 1 0; do
            echo "--- $spec fuse=$fuse"
            GGML_MI355X_FUSE=$fuse GGML_MI355X_DEBUG=1 timeout 600 tests/native/bin/model_parity "$m" 24 \
                > "$OUT/model_parity_$(basename "$m" .bin)_f$fuse.json" 2> "$OUT/model_parity_$(basename "$m" .bin)_f$fuse.err"
            echo "exit=$?"
            grep -E '"single"|"batch|"greedy"|encode_ms' "$OUT/model_parity_$(basename "$m" .bin)_f$fuse.json" | cut -c1-300
            grep -E "unsupported|failed|error" "$OUT/model_parity_$(basename "$m" .bin)_f$fuse.err" | sort | uniq -c | head -8
        done
    done
    ;;
bench)
    stage "whisper-bench (reference binary + plugin)"
    for spec in "base.en q5_0" "large-v3 q5_0"; do
        set -- $spec
        m=$(python3 scripts/synth_model.py --arch "$1" --qtype "$2") || continue
        echo "--- GPU $spec"
        timeout 900 "$REF/whisper-bench" -m "$m" -t 8 > "$OUT/bench_gpu_$1_$2.log" 2>&1
        grep -E "encode time|decode time|batchd time|prompt time|backends|MI355X" "$OUT/bench_gpu_$1_$2.log" | head -12
    done
    m=$(python3 scripts/synth_model.py --arch base.en --qtype q5_0)
    echo "--- CPU base.en q5_0 (-ng, $NCPU threads)"
    timeout 900 "$REF/whisper-bench" -m "$m" -ng -t "$NCPU" > "$OUT/bench_cpu_base.en_q5_0.log" 2>&1
    grep -E "encode time|decode time|batchd time|prompt time|system_info" "$OUT/bench_cpu_base.en_q5_0.log" | head -8
    ;;
prof)
    stage "per-kernel profile (hipEvents inside the backend)"
    m=$(python3 scripts/synth_model.py --arch large-v3 --qtype q5_0)
    GGML_MI355X_PROF=1 timeout 900 python3 bench.py --steps 1 --warmup 1 --profile-only > "$OUT/kernel_profile.json" 2> "$OUT/kernel_profile.err"
    tail -3 "$OUT/kernel_profile.err"; head -c 3000 "$OUT/kernel_profile.json"
    ;;
pytest)
    stage "pytest -m gpu"
    timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.txt"
    ;;
esac; done
echo; echo "=== done $(date +%T)"
