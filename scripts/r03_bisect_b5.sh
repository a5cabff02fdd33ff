#!/bin/bash
# which switch makes the 5-token batch of base.en Q5_0 -nfa / large-v3-turbo Q8_0 leave the reference?  (closing run of round 3: batch5 NMSE 2.5e-2 / 1.8e-1)
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out
export TMPDIR=/tmp GGML_MI355X_STRICT=1
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
mb=$(python3 scripts/synth_model.py --arch base.en --qtype q5_0)
run() { # label, model, steps, fa, env...
    local label=$1 m=$2 steps=$3 fa=$4; shift 4
    env "$@" timeout 600 tests/native/bin/model_parity "$m" "$steps" "$fa" > /tmp/mp.json 2> /tmp/mp.err || { echo "$label: rc=$?"; tail -3 /tmp/mp.err; return; }
    python3 -c "
import json; d=json.load(open('/tmp/mp.json')); print('$label'.ljust(58), 'single', d['single']['worst_nmse'], 'b5', d['batch5']['nmse'], 'b48', d['batch48']['nmse'])"
}
for rep in 1 2; do run "base.en nfa default (rep $rep)" "$mb" 16 0 X=1; done
run "base.en nfa VOCAB_KERNEL=0" "$mb" 16 0 GGML_MI355X_VOCAB_KERNEL=0
run "base.en nfa VOCAB_GROUPS=2" "$mb" 16 0 GGML_MI355X_VOCAB_GROUPS=2
run "base.en nfa LOGITS_MIRROR=0" "$mb" 16 0 GGML_MI355X_LOGITS_MIRROR=0
run "base.en nfa PLANES_MIN_T=99" "$mb" 16 0 GGML_MI355X_PLANES_MIN_T=99
run "base.en nfa GEMV_ROWS_MIN_T=3" "$mb" 16 0 GGML_MI355X_GEMV_ROWS_MIN_T=3
run "base.en FA default" "$mb" 16 1 X=1
if [ "${1:-}" = turbo ]; then
mt=$(python3 scripts/synth_model.py --arch large-v3-turbo --qtype q8_0)
run "turbo default" "$mt" 8 1 MODEL_PARITY_THREADS=32
run "turbo VOCAB_KERNEL=0" "$mt" 8 1 MODEL_PARITY_THREADS=32 GGML_MI355X_VOCAB_KERNEL=0
run "turbo LOGITS_MIRROR=0" "$mt" 8 1 MODEL_PARITY_THREADS=32 GGML_MI355X_LOGITS_MIRROR=0
fi
