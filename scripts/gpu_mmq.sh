#!/bin/bash
# One gpurun call around the int8 tile GEMM (csrc/kernels/mmq.hip): hardware probe, kernel tests, micro benchmark, encoder A-B, full suite.
# usage: scripts/gpu_mmq.sh [stage ...]   stages: probe mmqtest kbench encab sel pytest   (default: all; later stages are skipped when mmqtest fails)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
STAGES=${*:-probe mmqtest kbench encab pytest}
stage() { echo; echo "=== $1 === $(date +%T)"; }
ok=1
for s in $STAGES; do case $s in
probe)
    stage "hardware probe"
    timeout 60 scripts/probes/_bin/mmq_probe 2>&1 | tee "$OUT/mmq_probe.txt"
    ;;
mmqtest)
    stage "pytest tests/test_gpu_mmq.py + mul_mat_vs_oracle"
    timeout 900 python3 -m pytest tests/test_gpu_mmq.py tests/test_gpu.py -m gpu -q -p no:cacheprovider -k "mmq or rows or mul_mat_vs_oracle" > "$OUT/pytest_mmq.txt" 2>&1
    rc=$?; echo "exit=$rc"; tail -40 "$OUT/pytest_mmq.txt"
    [ $rc -ne 0 ] && ok=0
    ;;
kbench)
    [ $ok -eq 1 ] || { echo "skip kbench"; continue; }
    stage "mmq_kbench"
    { timeout 300 python3 scripts/mmq_kbench.py --qtype q5_0; timeout 200 python3 scripts/mmq_kbench.py --qtype q8_0 --what fc1,oproj; timeout 200 python3 scripts/mmq_kbench.py --qtype q4_k --what fc1,oproj;  timeout 200 python3 scripts/mmq_kbench.py --qtype q4_0 --what fc1; } > "$OUT/mmq_kbench.txt" 2>&1
    cat "$OUT/mmq_kbench.txt" | grep -v "^$" | tail -60
    ;;
encab)
    [ $ok -eq 1 ] || { echo "skip encab"; continue; }
    stage "bench.py encode A-B (mmq on / off)"
    for mmq in 1 0; do
        GGML_MI355X_MMQ=$mmq timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 > "$OUT/bench_mmq$mmq.json" 2> "$OUT/bench_mmq$mmq.err"
        echo "mmq=$mmq exit=$?"; python3 - "$OUT/bench_mmq$mmq.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","encode_ms","decode_ms_per_token","batchd_ms_per_token","prompt_ms_per_token")})
    print("roofline", d.get("roofline"))
    for k in d.get("kernels",[])[:12]: print("  ",k)
except Exception as e: print("parse failed", e)
PY
    done
    ;;
sel)
    stage "pytest selection: ${PYTEST_SEL:-}"
    timeout 1500 python3 -m pytest ${PYTEST_FILES:-tests} -m gpu -q -p no:cacheprovider -k "${PYTEST_SEL:-vad}" > "$OUT/pytest_sel.txt" 2>&1
    echo "exit=$?"; tail -25 "$OUT/pytest_sel.txt"
    ;;
pytest)
    stage "pytest -m gpu (everything)"
    timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.txt" 2>&1
    echo "exit=$?"; tail -30 "$OUT/pytest_gpu.txt"
    ;;
esac; done
echo; echo "=== done $(date +%T)"
