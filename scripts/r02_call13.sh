#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
timeout 900 python3 -m pytest tests/test_gpu.py -m gpu -q -p no:cacheprovider -x -k "flash_attn or layernorm or norm_gelu or op_parity or model_parity[large-v3-q5_0] or model_parity[base.en-q5_0] or model_parity[tiny.en or without_flash" 2>&1 | tail -5
run() {
  r=$(env "$@" timeout 300 python3 bench.py --steps 2 --warmup 1 --n-decode 4 --no-cpu-baseline --multi-stream 0 2>/dev/null | tail -1)
  python3 - "$r" "$*" <<'PY'
import json,sys
d=json.loads(sys.argv[1]); k=d.get("kernel_time_ms_per_chunk",{})
g=sum(v for n,v in k.items() if "ring_group" in n); s=sum(v for n,v in k.items() if "k_gemm_f16_ring<" in n)
print(f"{sys.argv[2]:40s} encode {d['encode_ms']:.3f} ms  prompt {d['prompt_ms_per_token']:.4f}  group {g:.3f} single {s:.3f} fattn {k.get('k_fattn_mfma(FattnArgs)',0):.3f} norm {k.get('k_norm_v4(NormArgs)',0):.3f} prep {k.get('k_prep_act(PrepArgs)',0):.3f}")
PY
}
for cfg in X=0 GGML_MI355X_FATTN_PREP_OUT=0 X=1 GGML_MI355X_FATTN_PREP_OUT=0; do run $cfg; done | tee $OUT/encoder_ab5.txt
date +%T
