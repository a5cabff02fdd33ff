mkdir -p gpurun_out/r05; export TMPDIR=/tmp
echo "== --gpus 2 on a one-GPU box"; python bench.py --gpus 2 --arch base.en --steps 1; echo "rc=$?"
echo "== pytest refuse"; timeout 300 python -m pytest tests/test_gpu.py -x -q -k "refuses_more_gpus" 2>&1 | tail -2
echo "== N=1 plain"; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err; echo "rc=$?"; tail -2 gpurun_out/r05/bench_n1.err
echo "== N=1 torchrun"; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 3 --warmup 1 --multi-stream 0 --no-cpu-baseline > gpurun_out/r05/bench_n1_torchrun.json 2> gpurun_out/r05/bench_n1_torchrun.err; echo "rc=$?"; tail -3 gpurun_out/r05/bench_n1_torchrun.err
python - <<'PY'
import json
for f in ("bench_n1", "bench_n1_torchrun"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["n_gpus"], d["encode_ms"], d["decode_ms_per_token"], d["batchd_ms_per_token"], d["prompt_ms_per_token"], d.get("weight_broadcast"), d.get("hip_runtime"))
        ms = d.get("multi_stream") or {}
        for k, v in ms.items():
            if isinstance(v, dict): print("   ", k, v.get("chunks_per_s"), v.get("decode_frac_of_hbm_peak"), v.get("error"))
    except Exception as e:
        print(f, "ERR", e)
PY
