#!/bin/bash
# usage: scripts/parity_bisect.sh <arch> <qtype> <steps> "ENV=a" ...  -> model_parity summary per environment (bisecting a logit difference)
cd "$(dirname "$0")/.."
ROOT=$PWD
arch=$1; qt=$2; steps=$3; shift 3
export GGML_MI355X_PLUGIN=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so GGML_BACKEND_PATH=$ROOT/whisper.cpp_amd/lib/libggml-mi355x.so
export LD_LIBRARY_PATH=$ROOT/oracle/_ref:$ROOT/whisper.cpp_amd/lib:${LD_LIBRARY_PATH:-}
export MODEL_PARITY_THREADS=${MODEL_PARITY_THREADS:-32}
m=$(python3 scripts/synth_model.py --arch "$arch" --qtype "$qt")
for v in "$@"; do
    env $v tests/native/bin/model_parity "$m" "$steps" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s single worst %.2e | batch5 %.2e | batch48 %.2e | greedy %d/%d' % (sys.argv[1], d['single']['worst_nmse'], d['batch5']['nmse'], d['batch48']['nmse'], d['greedy']['identical_prefix'], d['greedy']['steps']))" "$v"
done
