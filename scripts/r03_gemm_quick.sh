#!/bin/bash
# quick GPU check of a ring-GEMM change: bit-identity tests, scaling probe, encoder kernel table, whole encode
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu.py -q -m gpu --timeout 600 -k "ring or gemm or mul_mat_vs_oracle or f16_weight_copy" 2>&1 | tail -4
python3 scripts/gemm_probe.py 5120,1280,1500 5120,5120,1500 5120,1280,750 1280,1280,1500 1280,5120,1500 2560,1280,1500 2>&1 | grep -v amdgpu.ids
python3 scripts/enc_kbench.py --what fc1,fc2,oproj,qkv,xkv --iters 30 2>&1 | grep -v amdgpu.ids | grep "default"
python3 scripts/enc_ab.py --reps 8 -- "" "" 2>&1 | grep -v "^whisper_\|^ggml_\|load_backend"
