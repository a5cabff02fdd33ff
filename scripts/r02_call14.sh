#!/bin/bash
# the other BASELINE configurations through the bench protocol (headline config is large-v3 Q5_0)
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "large-v3 q4_k" "large-v3-turbo q8_0" "base.en q5_0" "tiny.en f16"; do
  set -- $cfg
  timeout 600 python3 bench.py --arch $1 --qtype $2 --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 > $OUT/bench_$1_$2.json 2> $OUT/bench_$1_$2.err
  python3 - $OUT/bench_$1_$2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['config']['workload'][:60], '| value', d['value'], 'encode', d['encode_ms'], 'decode', d['decode_ms_per_token'], 'batchd', d['batchd_ms_per_token'], 'prompt', d['prompt_ms_per_token'])
PY
done | tee $OUT/bench_other_configs.txt
