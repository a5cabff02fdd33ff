#!/bin/bash
# round-2 closing GPU call: full GPU suite, headline bench, rocprofv3 kernel trace + stats, PMC passes, stock whisper-bench + plugin
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
bash scripts/gpu_round.sh pytest bench prof pmc wbench > $OUT/round_full.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.txt | tail -8
cut -c1-700 $OUT/bench_large-v3_q5_0.json
grep -E "time =" $OUT/wbench_gpu_large-v3_q5_0.log
date +%T
