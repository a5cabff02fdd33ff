#!/bin/bash
# A-B measurements through bench.py: one line per environment setting (how profiles/r02_encoder_ab_*.txt, r02_decode_*_ab.txt were made).
#   usage: scripts/bench_ab.sh encode|decode [reps] -- "VAR=1" "VAR=0 OTHER=2" ...
# encode: 2 steps with 4 decodes each, per-kernel profile on (encode time from whisper's own timers + the GEMM / attention / LayerNorm
#         kernel totals per chunk);  decode: the full protocol, 3 steps, no profile (value, encode, decode, batchd, 4-stream throughput)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
what=$1; shift
reps=1; if [ "$1" != "--" ]; then reps=$1; shift; fi
shift
for i in $(seq $reps); do for cfg in "$@"; do
  if [ "$what" = encode ]; then args="--steps 2 --warmup 1 --n-decode 4 --no-cpu-baseline --multi-stream 0"; else args="--steps 3 --warmup 1 --no-cpu-baseline --multi-stream 4 --no-profile"; fi
  r=$(env $cfg timeout 600 python3 bench.py $args 2>/dev/null | tail -1)
  python3 - "$r" "$cfg" "$what" <<'PY'
import json, sys
d = json.loads(sys.argv[1]); k = d.get("kernel_time_ms_per_chunk") or {}
if sys.argv[3] == "encode":
    g = sum(v for n, v in k.items() if "ring_group" in n); s = sum(v for n, v in k.items() if "k_gemm_f16_ring<" in n)
    print(f"{sys.argv[2]:48s} encode {d['encode_ms']:.3f} ms  prompt {d['prompt_ms_per_token']:.4f}  group {g:.3f} single {s:.3f} "
          f"fattn {sum(v for n, v in k.items() if 'k_fattn_mfma' in n):.3f} norm {k.get('k_norm_v4(NormArgs)', 0):.3f} prep {k.get('k_prep_act(PrepArgs)', 0):.3f}")
else:
    h = d["hip_graph"]["host_ms_in_timed_region"]
    print(f"{sys.argv[2]:40s} value {d['value']:.2f} encode {d['encode_ms']:.3f} decode {d['decode_ms_per_token']:.4f} batchd {d['batchd_ms_per_token']:.4f} "
          f"4-stream {d['multi_stream']['chunks_per_s']:.3f} chunks/s  gpu_span/chunk {h['gpu_span'] / 3:.1f} set {h['set_tensor'] / 3:.2f} get {h['get_tensor'] / 3:.2f}")
PY
done; done
