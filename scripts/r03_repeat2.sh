#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp GGML_MI355X_STRICT=1 GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so REPEAT_VERBOSE=1
export LD_LIBRARY_PATH=$ROOT/whisper.cpp_amd/host/_whisper:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
mb=$(python3 scripts/synth_model.py --arch base.en --qtype q5_0)
echo "--- k_vocab + mirror, T=5, with MIRROR_CHECK"
GGML_MI355X_MIRROR_CHECK=1 timeout 300 tests/native/bin/repeat_check "$mb" 5 6 1 3 2>&1 | tail -12 | cut -c1-250
echo "--- k_vocab + mirror, T=5, without the check"
timeout 300 tests/native/bin/repeat_check "$mb" 5 6 1 3 2>&1 | tail -8 | cut -c1-250
echo "--- k_vocab, no mirror, T=5"
GGML_MI355X_LOGITS_MIRROR=0 timeout 300 tests/native/bin/repeat_check "$mb" 5 6 1 3 2>&1 | tail -8 | cut -c1-250
echo "--- k_gemv8 + mirror, T=5"
GGML_MI355X_VOCAB_KERNEL=0 timeout 300 tests/native/bin/repeat_check "$mb" 5 6 1 3 2>&1 | tail -8 | cut -c1-250
