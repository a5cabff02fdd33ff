mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q -k "attention_combine or multi_state or self_attention_straight" 2>&1 | tail -4
run() { echo "# $*"; env "$@" timeout 900 python3 scripts/stream_scaling.py --streams $S --batching 1 --steps 2 2>&1 | grep -v '^{"arch"' | cut -c1-200; }
{ S=16,32 run GGML_MI355X_MX_MIN_T=9
  S=16,32 run GGML_MI355X_ATTN_PLANES_MAX_KV=1536
} > gpurun_out/r05/mx_streams2.txt 2>&1
cat gpurun_out/r05/mx_streams2.txt
