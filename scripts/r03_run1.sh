#!/bin/bash
# Round 3, GPU call 1: new decode pipeline (kernel tests), cross-state batching (harness), XCD-masked streams, per-step host timeline,
# real-model search.  Every stage has its own timeout; everything lands in gpurun_out/r03_*.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
stage() { echo; echo "=== $1 === $(date +%T)"; }
STAGES=${*:-info probe ktests models harness xcd trace batchd}

for s in $STAGES; do case $s in
info)
    stage "box + real-model search (VERDICT r02 next #4c)"
    { rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; } > "$OUT/r03_info.txt" 2>&1
    { echo "--- files named ggml-*.bin / *.gguf / whisper model caches anywhere on the box:";
      timeout 60 find / -xdev \( -name "ggml-*.bin" -o -name "*.gguf" -o -name "for-tests-*.bin" \) -size +1M 2>/dev/null | grep -v "^$ROOT" | head -20;
      echo "--- ~/.cache, /models, /data, /mnt:"; ls -d /root/.cache/whisper* /root/.cache/huggingface /models /data /mnt/* 2>/dev/null | head;
      echo "--- network (2 s):"; timeout 4 python3 -c "import urllib.request; print(urllib.request.urlopen('https://huggingface.co/ggerganov/whisper.cpp/resolve/main/ggml-tiny.en.bin', timeout=2).status)" 2>&1 | tail -1; } > "$OUT/r03_real_model_search.txt" 2>&1
    cat "$OUT/r03_info.txt" "$OUT/r03_real_model_search.txt"
    ;;
probe)
    stage "CU-mask probe"
    timeout 120 scripts/_bin/cumask_probe > "$OUT/r03_cumask_probe.txt" 2>&1; echo "exit=$?"; cat "$OUT/r03_cumask_probe.txt"
    ;;
ktests)
    stage "pytest tests/test_gpu_batch.py (kernel-level part)"
    timeout 900 python3 -m pytest tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 180 --timeout-method=thread -k "not cross_state_batching" > "$OUT/r03_pytest_batch_kernels.txt" 2>&1
    echo "exit=$?"; tail -30 "$OUT/r03_pytest_batch_kernels.txt"
    ;;
models)
    stage "synthetic models"
    for spec in "base.en q5_0" "large-v3 q5_0" "large-v3-turbo q8_0"; do set -- $spec; python3 scripts/synth_model.py --arch "$1" --qtype "$2" > /dev/null; done
    ls -la /tmp/whisper_synth/ 2>/dev/null | tail -5
    ;;
harness)
    stage "cross-state batching through the harness"
    timeout 600 python3 -m pytest tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 300 --timeout-method=thread -k "cross_state_batching" > "$OUT/r03_pytest_batch_harness.txt" 2>&1
    echo "exit=$?"; tail -15 "$OUT/r03_pytest_batch_harness.txt"
    timeout 900 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 1,2,4,8 --batching 0,1 --n-decode 256 --steps 2 > "$OUT/r03_stream_scaling_batching.txt" 2>&1
    echo "exit=$?"; cat "$OUT/r03_stream_scaling_batching.txt"
    ;;
xcd)
    stage "one XCD-masked stream per state (both mask layouts)"
    for layout in 0 1; do
        GGML_MI355X_XCD_STREAMS=1 GGML_MI355X_XCD_MASK_LAYOUT=$layout timeout 600 python3 scripts/stream_scaling.py --arch large-v3 --qtype q5_0 --streams 1,4,8 --batching 0 --n-decode 256 --steps 2 \
            > "$OUT/r03_stream_scaling_xcd_layout$layout.txt" 2>&1
        echo "layout $layout exit=$?"; cat "$OUT/r03_stream_scaling_xcd_layout$layout.txt"
    done
    ;;
trace)
    stage "per-step host timeline (step_trace)"
    for spec in "large-v3 q5_0" "large-v3-turbo q8_0" "base.en q5_0"; do
        set -- $spec
        m=$(python3 scripts/synth_model.py --arch "$1" --qtype "$2")
        GGML_MI355X_TRACE=1 GGML_MI355X_STRICT=1 timeout 300 tests/native/bin/step_trace "$m" 256 32 > "$OUT/r03_step_trace_$1_$2.json" 2> "$OUT/r03_step_trace_$1_$2.err"
        echo "$spec exit=$?"; cat "$OUT/r03_step_trace_$1_$2.json"
    done
    ;;
batchd)
    stage "5-token steps: per-kernel profile + whisper timings"
    timeout 600 python3 bench.py --profile-only --profile-what batchd > "$OUT/r03_profile_batchd.json" 2> "$OUT/r03_profile_batchd.err"
    echo "exit=$?"; python3 - <<'E'
import json
d=json.load(open("gpurun_out/r03_profile_batchd.json"))
tot=sum(k["total_ms"] for k in d["kernels"])
print("total GPU ms over 16 steps:", round(tot,3), "-> per 5-token step", round(tot/16,4))
for k in d["kernels"][:14]: print(f'{k["name"][:70]:70s} {k["calls"]:6d} {k["total_ms"]*1e3/max(k["calls"],1):8.2f} us')
E
    GGML_MI355X_PLANES_MIN_T=99 timeout 600 python3 bench.py --profile-only --profile-what batchd > "$OUT/r03_profile_batchd_fused.json" 2> "$OUT/r03_profile_batchd_fused.err"
    python3 - <<'E'
import json
d=json.load(open("gpurun_out/r03_profile_batchd_fused.json"))
tot=sum(k["total_ms"] for k in d["kernels"])
print("(fused T=5 kernels of round 2) total GPU ms over 16 steps:", round(tot,3), "-> per step", round(tot/16,4))
E
    ;;
esac; done
echo; echo "=== done $(date +%T)"
