mkdir -p gpurun_out/r05; export TMPDIR=/tmp
b() { echo "# $*"; env "$@" timeout 600 python bench.py --steps 1 --warmup 1 --multi-stream 0 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['encode_ms'], d['decode_ms_per_token'], 'batchd', d['batchd_ms_per_token'], 'prompt', d['prompt_ms_per_token'])"; }
{ b GGML_MI355X_MX_LN_MIN_T=99
  b GGML_MI355X_MX_MIN_T=3 GGML_MI355X_MX_LN_MIN_T=99
  b GGML_MI355X_MX_MIN_T=3 GGML_MI355X_MX_LN_MIN_T=3 GGML_MI355X_LN_FUSED=1
  b GGML_MI355X_MX_LN_MIN_T=99 GGML_MI355X_LN_FUSED=1
} > gpurun_out/r05/beam_ab.txt 2>&1
cat gpurun_out/r05/beam_ab.txt
