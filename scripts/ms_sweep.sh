#!/bin/bash
# usage: scripts/ms_sweep.sh "ENV=a" ...  -> multi-stream throughput (bench.py --multi-stream $MS) per environment
cd "$(dirname "$0")/.."
for v in "$@"; do
    env $v python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --multi-stream ${MS:-4} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_stream']
print('%-44s single %.1f ms/chunk | %d streams: %.3f chunks/s, %.1f ms/chunk aggregate' % (sys.argv[1], d['value'], m['streams'], m.get('chunks_per_s', 0), m.get('ms_per_chunk_aggregate', 0)))" "$v"
done
