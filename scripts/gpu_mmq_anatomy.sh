#!/bin/bash
# anatomy of the mmq K-step (kernel with parts switched off) + PMC counters of the fc1 product
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== anatomy"; timeout 300 python3 scripts/mmq_kbench.py --qtype q5_0 --what fc1,oproj --anatomy --iters 20 2>&1 | grep -v amdgpu.ids | tee "$OUT/mmq_anatomy.txt"
timeout 200 python3 scripts/mmq_kbench.py --qtype q8_0 --what fc1 --anatomy --iters 20 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/mmq_anatomy.txt"
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_I8" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo "$ctr" | tr ' ' '+' | cut -c1-40)
  rm -rf "$OUT/pmc_mmq_$tag"
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace -f csv -d "$OUT/pmc_mmq_$tag" -o pmc -- python3 "$ROOT/scripts/mmq_kbench.py" --qtype q5_0 --what fc1 --iters 4 > "$OUT/pmc_mmq_$tag.log" 2>&1 )
  echo "--- $ctr exit=$?"
  python3 scripts/summarize_pmc.py "$OUT/pmc_mmq_$tag" 2>&1 | grep -i "mmq\|gemm_f16" | head -6
  find "$OUT/pmc_mmq_$tag" -name "*.csv" -size +20M -delete
done
echo "=== done"
