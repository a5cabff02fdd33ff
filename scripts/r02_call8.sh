#!/bin/bash
# grouped encoder GEMMs + LayerNorm/prep fusion + replica path on one GPU: full GPU suite, headline bench
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
bash scripts/gpu_round.sh pytest bench > $OUT/round_call8.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.txt | tail -12
cut -c1-900 $OUT/bench_large-v3_q5_0.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_large-v3_q5_0.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('kernel_time_ms_per_chunk'))[:1500])
PY
date +%T
