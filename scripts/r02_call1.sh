#!/bin/bash
# round-2 GPU call 1: where do kernel arguments live / what does a dependent launch cost, and does it move the decode step?
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{
for v in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; do
    echo "=== $v"; env $v timeout 120 scripts/_bin/launch_probe
done
} > $OUT/launch_probe.txt 2>&1
cat $OUT/launch_probe.txt
SWEEP_STEPS=2 SWEEP_ARGS="--multi-stream 0" timeout 900 scripts/env_sweep.sh "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" 2>&1 | tee $OUT/env_sweep_r02.txt
