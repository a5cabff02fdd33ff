#!/bin/bash
# usage: scripts/prof_what.sh <batchd|prompt|chunk> "ENV=a" ...  -> per-kernel hipEvent profile of that decode shape per environment
cd "$(dirname "$0")/.."
what=$1; shift
for v in "$@"; do
    echo "== $what [$v]"
    env $v python3 bench.py ${PROF_ARGS:-} --profile-only --profile-what $what 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); tot=sum(k['total_ms'] for k in d['kernels']); print('total ms', round(tot,3))
for k in d['kernels'][:int('${PROF_TOP:-8}')]: print('%-58s calls %6d avg_us %8.2f total_ms %8.3f GB/s %7.1f' % (k['name'][:58], k['calls'], k['total_ms']*1e3/k['calls'], k['total_ms'], k['algo_bytes']/k['total_ms']/1e6 if k['total_ms'] else 0))"
done
