#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python3 -m pytest tests/test_gpu.py -m gpu -q -p no:cacheprovider -k "graph_replay" 2>&1 | tail -4 | cut -c1-300
