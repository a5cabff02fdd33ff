#!/bin/bash
# Closing run of a round: full GPU suite, bench (headline + roofline + cpu_baseline), rocprofv3 kernel stats, PMC passes, the stock whisper-bench,
# the other BASELINE configurations.   usage: scripts/gpu_final.sh
cd "$(dirname "$0")/.."
export ROUND=${ROUND:-5}
bash scripts/gpu_round.sh ${FINAL_STAGES:-pytest bench prof pmc wbench}
OUT=gpurun_out
# BASELINE.json configs[2]: large-v3 Q4_K with HBM / MFMA counters (VERDICT r04 missing #5)
BENCH_ARCH=large-v3 BENCH_QTYPE=q4_k bash scripts/gpu_round.sh pmc 2>&1 | tail -40
# merged decode chains: stream scaling (default widths) and the rocprofv3 anatomy of a 16- and a 32-column chain step (decode_mx.hip)
mkdir -p $OUT/r05
{ echo "# default chain widths, matrix-core mat-vecs from 9 columns"; timeout 900 python3 scripts/stream_scaling.py --streams 6,8,12,16,24,32 --batching 1 --steps 2 2>&1 | grep -v '^{"arch"' | cut -c1-240
  echo "# GGML_MI355X_MX_MIN_T=0 (k_gemv_q / k_vocab)"; GGML_MI355X_MX_MIN_T=0 timeout 900 python3 scripts/stream_scaling.py --streams 16,32 --batching 1 --steps 2 2>&1 | grep -v '^{"arch"' | cut -c1-240
} > $OUT/r05/stream_scaling.txt 2>&1
cat $OUT/r05/stream_scaling.txt
bash scripts/mx_trace.sh 2>&1 | tail -60
GGML_MI355X_MX_MIN_T=9 timeout 300 python3 scripts/mx_kbench.py --T 8,12,16,32 > $OUT/r05/mx_kbench_final.json 2> /dev/null
echo; echo "=== other BASELINE configurations === $(date +%T)"
{ for cfg in "large-v3-turbo q8_0" "large-v3 q4_k" "base.en q5_0" "tiny.en f16"; do set -- $cfg
    echo "# $1 $2"; timeout 600 python3 bench.py --arch $1 --qtype $2 --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','encode_ms','decode_ms_per_token','batchd_ms_per_token','prompt_ms_per_token')}, 'roofline', {k:d['roofline'].get(k) for k in ('kernel','frac','step_frac','encode_frac')})"
  done; } > $OUT/bench_other_configs.txt 2>&1
cat $OUT/bench_other_configs.txt
echo "=== done $(date +%T)"
