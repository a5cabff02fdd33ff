#!/bin/bash
# rocprofv3 kernel durations of the stand-alone probes: vocabulary projection (scripts/logits_bench.py) and the read-once stream ceiling (scripts/probes/stream_probe.hip)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
summ() { python3 - "$1" "$2" <<'P'
import csv, glob, collections, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not fs: print("no trace under", sys.argv[1]); sys.exit(0)
d = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:44], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v.sort(); print(k, "launches", len(v), "median us", v[len(v)//2] / 1e3, "min", v[0] / 1e3, "-> TB/s at 45.64 MB", round(45.64208e6 / (v[len(v)//2] * 1e-9) / 1e12, 2))
P
}
# CFGS: space-separated settings, each NAME=VALUE[,NAME=VALUE...]
for cfg in ${CFGS:-GGML_MI355X_VOCAB_GROUPS=2 GGML_MI355X_VOCAB_KERNEL=0,GGML_MI355X_GEMV_PASS_WAVES=16}; do
    ( cd /tmp && env $(echo "$cfg" | tr ',' ' ') timeout 200 rocprofv3 --kernel-trace -f csv -d "$OUT/prof_logits_$cfg" -o l -- python3 "$ROOT/scripts/logits_bench.py" ${LB_ARGS:-} > "$OUT/prof_logits_$cfg.json" 2> "$OUT/prof_logits_$cfg.err" )
    echo "logits $cfg ${LB_ARGS:-}: $(tail -1 "$OUT/prof_logits_$cfg.json" | cut -c1-120)"; summ "$OUT/prof_logits_$cfg" "k_gemv8"; summ "$OUT/prof_logits_$cfg" "k_vocab"
done
if [ -z "${NO_STREAM:-}" ]; then
( cd /tmp && timeout 200 rocprofv3 --kernel-trace -f csv -d "$OUT/prof_stream" -o s -- "$ROOT/scripts/probes/_bin/stream_probe" > "$OUT/prof_stream.txt" 2> "$OUT/prof_stream.err" )
summ "$OUT/prof_stream" k_read
fi
rm -rf "$OUT"/prof_logits_*/ "$OUT"/prof_stream/
