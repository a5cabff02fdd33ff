#!/bin/bash
# usage: scripts/env_sweep.sh "ENV=a ENV2=b" "ENV=c" ...   -> one short bench.py run per environment, key numbers per line
cd "$(dirname "$0")/.."
for v in "$@"; do
    env $v python3 bench.py --steps ${SWEEP_STEPS:-4} --warmup 1 --no-cpu-baseline --no-profile ${SWEEP_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hip_graph']['host_ms_in_timed_region']
print('%-48s value %.1f enc %.2f dec %.4f batchd %.4f gpu_span/step %.1f plan %.1f patch %.1f launch %.1f' % (sys.argv[1], d['value'], d['encode_ms'], d['decode_ms_per_token'], d['batchd_ms_per_token'], h['gpu_span']/d['steps'], h['plan'], h['patch'], h['launch']))" "$v"
done
