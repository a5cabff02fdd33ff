#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python3 bench.py --profile-only --profile-what batchd > $OUT/profile_batchd.txt 2>&1; tail -40 $OUT/profile_batchd.txt
