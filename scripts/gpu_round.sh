#!/bin/bash
# One gpurun call: GPU pytest, bench, rocprofv3 kernel trace (+ optional PMC passes).  Logs land in gpurun_out/.
# usage: scripts/gpu_round.sh [stage ...]    stages: pytest bench prof pmc wbench kbench   (default: pytest bench prof)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
STAGES=${*:-pytest bench prof}
ARCH=${BENCH_ARCH:-large-v3}
QT=${BENCH_QTYPE:-q5_0}
stage() { echo; echo "=== $1 === $(date +%T)"; }

{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; } > "$OUT/info.txt" 2>&1

for s in $STAGES; do case $s in
pytest)
    stage "pytest -m gpu"
    timeout 1700 python3 -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.txt" 2>&1
    echo "exit=$?"; tail -40 "$OUT/pytest_gpu.txt"
    ;;
bench)
    stage "bench.py $ARCH $QT"
    timeout 1500 python3 bench.py --arch "$ARCH" --qtype "$QT" > "$OUT/bench_${ARCH}_${QT}.json" 2> "$OUT/bench_${ARCH}_${QT}.err"
    echo "exit=$?"; tail -3 "$OUT/bench_${ARCH}_${QT}.err"; cat "$OUT/bench_${ARCH}_${QT}.json"
    ;;
prof)
    stage "rocprofv3 --kernel-trace --stats (bench.py $ARCH $QT, 1 step)"
    rm -rf "$OUT/prof"
    ( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof" -o bench -- python3 "$ROOT/bench.py" --arch "$ARCH" --qtype "$QT" --steps 1 --warmup 1 --no-cpu-baseline --multi-stream 0 \
        > "$OUT/prof_bench.json" 2> "$OUT/prof_bench.err" )
    echo "exit=$?"
    f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats_${ARCH}_${QT}.csv"; head -25 "$f"; fi
    python3 scripts/summarize_trace.py "$OUT/prof" > "$OUT/kernel_trace_summary_${ARCH}_${QT}.txt" 2>&1; grep -v "at::\|rocclr" "$OUT/kernel_trace_summary_${ARCH}_${QT}.txt" | head -24; sed -n '/decode-step anatomy/,$p' "$OUT/kernel_trace_summary_${ARCH}_${QT}.txt" | head -50
    # the raw trace is large: keep only the stats
    find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
    ;;
pmc)
    stage "rocprofv3 --pmc passes (bench.py $ARCH $QT, decode only)"
    for ctr in ${PMC_SETS:-"FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"}; do
        tag=$(echo "$ctr" | tr ' ' '+')
        rm -rf "$OUT/pmc_$tag"
        ( cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace -f csv -d "$OUT/pmc_$tag" -o pmc -- python3 "$ROOT/bench.py" --arch "$ARCH" --qtype "$QT" --steps 1 --warmup 0 --n-decode 8 --no-cpu-baseline --no-profile --multi-stream 0 \
            > "$OUT/pmc_$tag.json" 2> "$OUT/pmc_$tag.err" )
        echo "$tag exit=$?"
        python3 scripts/summarize_pmc.py "$OUT/pmc_$tag" > "$OUT/pmc_$tag.summary.txt" 2>&1; head -30 "$OUT/pmc_$tag.summary.txt"
    done
    # per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of THIS run (bench.py's roofline.traffic reads profiles/pmc_traffic.json)
    # (one entry per configuration: merged into the committed file's other configurations)
    python3 scripts/make_pmc_traffic.py "$OUT" ${ROUND:-5} "$ARCH $QT" "$ROOT/profiles/pmc_traffic.json" > "$OUT/pmc_traffic_${ARCH}_${QT}.json" 2> "$OUT/pmc_traffic.err"; head -c 400 "$OUT/pmc_traffic_${ARCH}_${QT}.json"; echo
    for t in FETCH_SIZE WRITE_SIZE; do rm -rf "$OUT/pmc_${ARCH}_${QT}_$t"; mv "$OUT/pmc_$t" "$OUT/pmc_${ARCH}_${QT}_$t" 2>/dev/null; done
    for d in "$OUT"/pmc_*/; do find "$d" -name "*.csv" -size +20M -delete; done
    ;;
kbench)
    stage "kernel micro-benchmarks under rocprofv3 (variants: $KB_VARIANTS)"
    i=0
    for var in ${KB_VARIANTS:-default}; do
        i=$((i+1)); tag="kb$i"
        rm -rf "$OUT/$tag"
        ( cd /tmp && env $(echo "$var" | tr ',' ' ' | sed 's/^default$//') timeout 300 rocprofv3 --kernel-trace -f csv -d "$OUT/$tag" -o kb -- python3 "$ROOT/scripts/kbench.py" ${KB_ARGS:-} > "$OUT/$tag.json" 2> "$OUT/$tag.err" )
        echo "--- $tag [$var] exit=$?"
        python3 scripts/summarize_trace.py "$OUT/$tag" > "$OUT/$tag.trace.txt" 2>&1
        echo "variant: $var" >> "$OUT/$tag.trace.txt"
        head -30 "$OUT/$tag.trace.txt"
        find "$OUT/$tag" -name "*.csv" -size +8M -delete
    done
    ;;
wbench)
    stage "reference whisper-bench binary + plugin"
    export GGML_BACKEND_PATH=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
    export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
    m=$(python3 scripts/synth_model.py --arch "$ARCH" --qtype "$QT")
    timeout 900 whisper.cpp_amd/host/_whisper/whisper-bench -m "$m" -t 8 > "$OUT/wbench_gpu_${ARCH}_${QT}.log" 2>&1
    grep -E "encode time|decode time|batchd time|prompt time|backends|MI355X" "$OUT/wbench_gpu_${ARCH}_${QT}.log" | head -12
    ;;
esac; done
echo; echo "=== done $(date +%T)"
