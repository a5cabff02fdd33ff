#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
date +%T
timeout 2700 python3 -m pytest tests -m gpu -q -p no:cacheprovider -rf -s > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit=$?"; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.txt | head -30
grep -E "exact attention|mi355x_log_mel:" $OUT/pytest_gpu.txt | head -20
date +%T
# step block: correctness of the opt-in path (agent-scope loads) on one model
M=$(python3 scripts/synth_model.py --arch base.en --qtype q5_0)
GGML_MI355X_STEP_BLOCK=1 GGML_MI355X_STRICT=1 GGML_MI355X_PLUGIN=$PWD/whisper.cpp_amd/lib/libggml-mi355x.so LD_LIBRARY_PATH=$PWD/oracle/_ref:$PWD/whisper.cpp_amd/lib \
  timeout 600 tests/native/bin/model_parity $M 64 > $OUT/model_parity_base.en_q5_0_stepblock.json 2>$OUT/stepblock.err
python3 -c "
import json; d=json.load(open('$OUT/model_parity_base.en_q5_0_stepblock.json')); s=d['single']
print('STEP_BLOCK=1 base.en q5_0: worst %.2e mean %.2e agree %d/%d b5 %.2e greedy %d' % (s['worst_nmse'], s['mean_nmse'], s['argmax_agree'], s['steps'], d['batch5']['nmse'], d['greedy']['identical_prefix']))"
date +%T
bash scripts/gpu_round.sh bench prof > $OUT/round_bench_prof.log 2>&1; tail -5 $OUT/round_bench_prof.log | cut -c1-300
date +%T
for q in 4 16; do STREAM_HW_QUEUES=$q GPU_MAX_HW_QUEUES=$q timeout 600 python3 scripts/stream_scaling.py large-v3 q5_0 3 4 5 6 2>&1 | tail -1 | cut -c1-600; done | tee $OUT/stream_scaling_queues.txt
date +%T
