#!/bin/bash
# encoder A-B: grouped GEMM tile width / ring depth, grouping off, LayerNorm+prep fusion off (encode_ms from whisper's own timings)
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  r=$(env "$@" timeout 300 python3 bench.py --steps 2 --warmup 1 --n-decode 4 --no-cpu-baseline --multi-stream 0 2>/dev/null | tail -1)
  python3 - "$r" "$*" <<'PY'
import json,sys
d=json.loads(sys.argv[1]); k=d.get("kernel_time_ms_per_chunk",{})
g=sum(v for n,v in k.items() if "ring_group" in n); s=sum(v for n,v in k.items() if "k_gemm_f16_ring<" in n)
print(f"{sys.argv[2]:48s} encode {d['encode_ms']:.3f} ms  prompt {d['prompt_ms_per_token']:.4f}  group {g:.3f} ms single {s:.3f} ms  norm {k.get('k_norm_v4(NormArgs)',0):.3f} prep {k.get('k_prep_act(PrepArgs)',0):.3f}")
PY
}
date +%T
for cfg in GGML_MI355X_GEMM_GROUP_CFG=642 "GGML_MI355X_GEMM_GROUP_CFG=642 GGML_MI355X_GEMM_RING_BIG=642" "GGML_MI355X_GEMM_GROUP_CFG=642 GGML_MI355X_GEMM_RING_BIG=643" "GGML_MI355X_GEMM_GROUP_CFG=642 GGML_MI355X_GEMM_RING_NST64=2" "GGML_MI355X_GEMM_GROUP_CFG=642 GGML_MI355X_GEMM_RING_NST64=3" "GGML_MI355X_GEMM_GROUP_CFG=642 GGML_MI355X_GEMM_RING_NST64=5" GGML_MI355X_GEMM_GROUP_CFG=642 GGML_MI355X_GEMM_GROUP_CFG=1282; do
  run $cfg
done | tee $OUT/encoder_ab2.txt
date +%T
