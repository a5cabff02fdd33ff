#!/bin/bash
# round-2 GPU call 2: the whole GPU test suite (all BASELINE configs, exact mode, STRICT op parity, -nfa) and the headline bench
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
timeout 2700 python3 -m pytest tests -m gpu -q -p no:cacheprovider -rf ${PYTEST_ARGS:-} > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit=$?"; tail -60 $OUT/pytest_gpu.txt
date +%T
if [ -z "${SKIP_BENCH:-}" ]; then
timeout 900 python3 bench.py > $OUT/bench_large-v3_q5_0.json 2> $OUT/bench_large-v3_q5_0.err
echo "bench exit=$?"; tail -3 $OUT/bench_large-v3_q5_0.err; cut -c1-1500 $OUT/bench_large-v3_q5_0.json
fi
date +%T
