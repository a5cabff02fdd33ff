#!/bin/bash
# rocprofv3 kernel trace of merged chains (one chain as wide as the states): per-kernel time per chain step
cd "$(dirname "$0")/.." 2>/dev/null || cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
for S in ${STREAMS:-16 32}; do
  rm -rf /tmp/tr$S
  GGML_MI355X_BATCH_COLS=${COLS:-32} timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/tr$S -- python3 scripts/stream_scaling.py --streams $S --batching 1 --steps 1 --n-decode 96 > $OUT/trace${S}_run.txt 2>&1
  tail -2 $OUT/trace${S}_run.txt | cut -c1-200
  python3 scripts/chain_anatomy.py /tmp/tr$S > $OUT/chain_anatomy_${S}cols.txt 2>&1
  cat $OUT/chain_anatomy_${S}cols.txt
done
