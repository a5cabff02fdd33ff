#!/bin/bash
# GPU check of the one-launch self-attention block: kernel test, whole-model parity, A-B of the headline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python3 -m pytest tests/test_gpu_head.py -x -q -m gpu --timeout 600 2>&1 | tail -12
for v in 1 0 1 0; do
  r=$(GGML_MI355X_SELF_HEAD=$v timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 --no-profile 2>/dev/null | tail -1)
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); h=d['backend']['host_ms_in_timed_region'] if 'backend' in d else {}
print('SELF_HEAD=$v', 'ms/chunk', d['value'], 'encode', d['encode_ms'], 'decode ms/token', d['decode_ms_per_token'], 'batchd', d['batchd_ms_per_token'], 'gpu_span per chunk', round(h.get('gpu_span', 0)/3, 1))" "$r"
done
timeout 1200 python3 -m pytest tests/test_gpu.py -x -q -m gpu --timeout 1000 -k "model_parity or whisper_full or first_multi_token" 2>&1 | tail -6
