#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
timeout 900 python3 -m pytest tests/test_gpu.py -m gpu -q -p no:cacheprovider -x -k "concurrent_streams or native_harness or payload_skipping or model_parity[base.en-q5_0] or model_parity[large-v3-q5_0] or whisper_full_pipeline[micro or bench_smoke" 2>&1 | tail -5
run() {
  r=$(env "$@" timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 4 --no-profile 2>/dev/null | tail -1)
  python3 - "$r" "$*" <<'PY'
import json,sys
d=json.loads(sys.argv[1])
print(f"{sys.argv[2]:28s} value {d['value']:.2f} encode {d['encode_ms']:.3f} decode {d['decode_ms_per_token']:.4f} batchd {d['batchd_ms_per_token']:.4f} 4-stream {d['multi_stream']['chunks_per_s']:.3f} chunks/s  gpu_span/chunk {d['hip_graph']['host_ms_in_timed_region']['gpu_span']/3:.1f} set {d['hip_graph']['host_ms_in_timed_region']['set_tensor']/3:.2f} get {d['hip_graph']['host_ms_in_timed_region']['get_tensor']/3:.2f} sync {d['hip_graph']['host_ms_in_timed_region']['synchronize']/3:.1f}")
PY
}
for i in 1 2 3; do for cfg in GGML_MI355X_PREFETCH_OUT=1 GGML_MI355X_PREFETCH_OUT=0; do run $cfg; done; done | tee $OUT/prefetch_ab.txt
date +%T
