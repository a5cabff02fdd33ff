mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_mmq.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "plane or concurrent_streams or first_multi_token" 2>&1 | tail -4
