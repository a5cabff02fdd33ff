#!/bin/bash
# determinism of multi-token steps under the plugin's switches (tests/native/repeat_check.cpp)
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp GGML_MI355X_STRICT=1 GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
mb=$(python3 scripts/synth_model.py --arch base.en --qtype q5_0)
run() { local label=$1; shift; echo -n "$label: "; env "$@" timeout 300 tests/native/bin/repeat_check $ARGS 2>&1 | tail -1; }
ARGS="$mb 5 60 1 3"
run "base.en FA T=5 default" X=1
run "base.en FA T=5 VOCAB_KERNEL=0" GGML_MI355X_VOCAB_KERNEL=0
run "base.en FA T=5 PLANES_MIN_T=99" GGML_MI355X_PLANES_MIN_T=99
run "base.en FA T=5 LOGITS_MIRROR=0" GGML_MI355X_LOGITS_MIRROR=0
run "base.en FA T=5 SELF_ATTN_PLANES=0" GGML_MI355X_SELF_ATTN_PLANES=0
run "base.en FA T=5 POUT_ROWS=2" GGML_MI355X_POUT_ROWS=2
run "base.en FA T=5 VOCAB_KERNEL=0+MIRROR=0" GGML_MI355X_VOCAB_KERNEL=0 GGML_MI355X_LOGITS_MIRROR=0
ARGS="$mb 5 60 0 3"; run "base.en nfa T=5 default" X=1
ARGS="$mb 3 60 1 3"; run "base.en FA T=3 default" X=1
ARGS="$mb 8 60 1 3"; run "base.en FA T=8 default" X=1
ARGS="$mb 2 60 1 3"; run "base.en FA T=2 default" X=1
ARGS="$mb 1 60 1 3"; run "base.en FA T=1 default" X=1
ARGS="$mb 48 30 1 3"; run "base.en FA T=48 default" X=1
