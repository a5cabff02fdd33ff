#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider -rf -s -k "torchrun or log_mel or native_harness or language or step_block" > $OUT/pytest_gpu_subset.txt 2>&1
echo "pytest exit=$?"; grep -E "passed|failed" $OUT/pytest_gpu_subset.txt | tail -2; grep -E "^FAILED|^ERROR|mi355x_log_mel:|bench.py: rank" $OUT/pytest_gpu_subset.txt | head
date +%T
M=$(python3 scripts/synth_model.py --arch base.en --qtype q5_0)
GGML_MI355X_STEP_BLOCK=1 GGML_MI355X_STRICT=1 GGML_MI355X_PLUGIN=$PWD/whisper.cpp_amd/lib/libggml-mi355x.so LD_LIBRARY_PATH=$PWD/oracle/_ref:$PWD/whisper.cpp_amd/lib \
  timeout 600 tests/native/bin/model_parity $M 64 > $OUT/model_parity_base.en_q5_0_stepblock.json 2>$OUT/stepblock.err
python3 -c "
import json; d=json.load(open('$OUT/model_parity_base.en_q5_0_stepblock.json')); s=d['single']
print('STEP_BLOCK=1 base.en q5_0: worst %.2e mean %.2e agree %d/%d b5 %.2e greedy %d' % (s['worst_nmse'], s['mean_nmse'], s['argmax_agree'], s['steps'], d['batch5']['nmse'], d['greedy']['identical_prefix']))"
date +%T
timeout 120 scripts/_bin/persist_probe 32 2>&1 | tee $OUT/persist_probe.txt
timeout 120 scripts/_bin/persist_probe 4 2>&1 | tail -2 | tee -a $OUT/persist_probe.txt
date +%T
for q in 2 3 4; do STREAM_HW_QUEUES=$q GPU_MAX_HW_QUEUES=$q timeout 600 python3 scripts/stream_scaling.py large-v3 q5_0 6 8 12 16 2>&1 | tail -1 | cut -c1-700; done | tee $OUT/stream_scaling_queues2.txt
date +%T
