#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  r=$(env "$@" timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 --no-profile 2>/dev/null | tail -1)
  python3 - "$r" "$*" <<'PY'
import json,sys
d=json.loads(sys.argv[1])
print(f"{sys.argv[2]:40s} value {d['value']:.2f} encode {d['encode_ms']:.3f} decode {d['decode_ms_per_token']:.4f} batchd {d['batchd_ms_per_token']:.4f}")
PY
}
for i in 1 2; do for cfg in GGML_MI355X_SPAN=1 GGML_MI355X_SPAN=0; do run $cfg; done; done | tee $OUT/span_ab.txt
