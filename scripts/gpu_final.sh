#!/bin/bash
# Closing run of a round: full GPU suite, bench (headline + roofline + cpu_baseline), rocprofv3 kernel stats, PMC passes, the stock whisper-bench,
# the other BASELINE configurations.   usage: scripts/gpu_final.sh
cd "$(dirname "$0")/.."
export ROUND=${ROUND:-4}
bash scripts/gpu_round.sh pytest bench prof pmc wbench
OUT=gpurun_out
echo; echo "=== other BASELINE configurations === $(date +%T)"
{ for cfg in "large-v3-turbo q8_0" "large-v3 q4_k" "base.en q5_0" "tiny.en f16"; do set -- $cfg
    echo "# $1 $2"; timeout 600 python3 bench.py --arch $1 --qtype $2 --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','encode_ms','decode_ms_per_token','batchd_ms_per_token','prompt_ms_per_token')}, 'roofline', {k:d['roofline'].get(k) for k in ('kernel','frac','step_frac','encode_frac')})"
  done; } > $OUT/bench_other_configs.txt 2>&1
cat $OUT/bench_other_configs.txt
echo "=== done $(date +%T)"
