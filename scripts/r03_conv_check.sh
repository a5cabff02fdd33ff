#!/bin/bash
# GPU check of the conv front-end fusion + transpose kernel + attention default: kernel tests, op parity through the plugin, two whole models, encode timing
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu --timeout 600 -k "transposing or bias_per or key_groups" 2>&1 | tail -4
timeout 1200 python3 -m pytest tests/test_gpu.py -x -q -m gpu --timeout 1000 -k "op_parity or model_parity or cpy or full_size or whisper_full" 2>&1 | tail -6
python3 scripts/enc_ab.py --reps 8 -- "" "" 2>&1 | grep -v "^whisper_\|^ggml_\|load_backend"
python3 scripts/enc_ab.py --arch base.en --qtype q5_0 --reps 8 -- "" 2>&1 | grep -v "^whisper_\|^ggml_\|load_backend"
