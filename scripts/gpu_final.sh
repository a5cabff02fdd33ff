#!/bin/bash
# Closing run of a round: full GPU suite, bench (headline + roofline + cpu_baseline), rocprofv3 kernel stats, PMC passes, the stock whisper-bench,
# the other BASELINE configurations.   usage: scripts/gpu_final.sh
cd "$(dirname "$0")/.."
export ROUND=${ROUND:-6}
bash scripts/gpu_round.sh ${FINAL_STAGES:-pytest bench prof pmc wbench}
OUT=gpurun_out
# BASELINE.json configs[2]: large-v3 Q4_K with HBM / MFMA counters (VERDICT r04 missing #5)
BENCH_ARCH=large-v3 BENCH_QTYPE=q4_k bash scripts/gpu_round.sh pmc 2>&1 | tail -40
# merged decode chains: stream scaling (default widths) and the rocprofv3 anatomy of a 16- and a 32-column chain step (decode_mx.hip)
mkdir -p $OUT/r06
{ echo "# default chain widths (60 % rule up to 32 decoding states, equal widths beyond), matrix-core mat-vecs from 9 columns"; timeout 1500 python3 scripts/stream_scaling.py --streams 6,8,12,16,24,32,48,64 --batching 1 --steps 2 2>&1 | grep -v '^{"arch"' | cut -c1-240
} > $OUT/r06/stream_scaling.txt 2>&1
cat $OUT/r06/stream_scaling.txt
# MFMA utilisation as a percentage (VERDICT r05 missing #4): one pass with the MFMA counters AND GRBM_GUI_ACTIVE, default policy and int8-everywhere, Q5_0 and Q4_K
for cfg in "q5_0 1" "q4_k 1" "q5_0 2" "q4_k 2"; do set -- $cfg; q=$1; m=$2
  rm -rf $OUT/pmc_mfma_${q}_mmq$m
  ( cd /tmp && GGML_MI355X_MMQ=$m timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OLDPWD/$OUT/pmc_mfma_${q}_mmq$m -o pmc -- python3 $OLDPWD/bench.py --qtype $q --steps 1 --warmup 1 --n-decode 4 --no-cpu-baseline --no-profile --multi-stream 0 > $OLDPWD/$OUT/pmc_mfma_${q}_mmq$m.json 2> $OLDPWD/$OUT/pmc_mfma_${q}_mmq$m.err )
  { echo "# large-v3 $q, GGML_MI355X_MMQ=$m (1: f16 ring from 1024 columns on, int8 tile GEMM below; 2: int8 tile GEMM at every width), bench.py --n-decode 4 under rocprofv3 --pmc"; python3 scripts/mfma_util.py $OUT/pmc_mfma_${q}_mmq$m; } > $OUT/r06/mfma_util_${q}_mmq$m.txt 2>&1
  head -12 $OUT/r06/mfma_util_${q}_mmq$m.txt | cut -c1-250
  find $OUT/pmc_mfma_${q}_mmq$m -name "*.csv" -size +8M -delete
done
echo; echo "=== other BASELINE configurations === $(date +%T)"
{ for cfg in "large-v3-turbo q8_0" "large-v3 q4_k" "base.en q5_0" "tiny.en f16"; do set -- $cfg
    echo "# $1 $2"; timeout 600 python3 bench.py --arch $1 --qtype $2 --steps 3 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','encode_ms','decode_ms_per_token','batchd_ms_per_token','prompt_ms_per_token')}, 'roofline', {k:d['roofline'].get(k) for k in ('kernel','frac','step_frac','encode_frac')})"
  done; } > $OUT/bench_other_configs.txt 2>&1
cat $OUT/bench_other_configs.txt
echo "=== done $(date +%T)"
