#!/bin/bash
# attribute the round-3 -> round-4 slow-downs of the non-headline configurations.  Same box, same call:
# the closing commit of round 3 (worktree _r03b, built beside HEAD) against HEAD, HEAD with batching off, HEAD with the f16 ring instead of the int8 tile GEMM.
cd "$(dirname "$0")/.." 2>/dev/null || cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
line() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%8.2f ms/chunk  encode %6.3f  decode %.4f  batchd %.4f  prompt %.4f' % (d['value'], d['encode_ms'], d['decode_ms_per_token'], d['batchd_ms_per_token'], d['prompt_ms_per_token']))"; }
b() { dir=$1; shift; label=$1; shift; printf "%-34s" "$label"; ( cd $dir && env "$@" timeout 600 python3 bench.py --arch $ARCH --qtype $QT --steps 3 --warmup 1 --multi-stream 0 --no-cpu-baseline 2>/dev/null | line ); }
{
for cfg in "base.en q5_0" "tiny.en f16" "large-v3-turbo q8_0" "large-v3 q4_k" "large-v3 q5_0"; do set -- $cfg; ARCH=$1; QT=$2
  echo "## $ARCH $QT"
  b _r03b "r03b (81be2d8)" X=1
  b .     "HEAD" X=1
  b .     "HEAD GGML_MI355X_BATCH=0" GGML_MI355X_BATCH=0
  b .     "HEAD GGML_MI355X_MMQ=0" GGML_MI355X_MMQ=0
  b _r03b "r03b again" X=1
  b .     "HEAD again" X=1
done
} > $OUT/regress_ab.txt 2>&1
cat $OUT/regress_ab.txt
