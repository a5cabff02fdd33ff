"""Host-side logic that needs no GPU: planar re-layout, GELU table, synthetic model files, roofline arithmetic,
and the multi-process (world_size 2, gloo) path of the benchmark harness."""
import ctypes as C
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, ptr

from whisper_cpp_amd import kernels_api as ka

G = Path(__file__).resolve().parent / "golden"
BLK = {ka.Q4_0: (32, 18), ka.Q5_0: (32, 22), ka.Q8_0: (32, 34), ka.Q4_K: (256, 144)}


@pytest.mark.parametrize("t", list(BLK))
def test_planar_roundtrip_and_layout(t):
    per, size = BLK[t]
    nb = 37
    rng = np.random.default_rng(t)
    blocks = rng.integers(0, 256, nb * size, dtype=np.uint8)
    planar = ka.repack_to_planar(t, blocks, nb * per)
    assert planar.size == blocks.size                      # same byte count as ggml's layout
    assert np.array_equal(ka.repack_from_planar(t, planar, nb * per), blocks)
    b = blocks.reshape(nb, size)
    if t == ka.Q5_0:      # ggml block: d(2) qh(4) qs(16)  ->  planes qs | qh | d   (include/mi355x_kernels.h)
        assert np.array_equal(planar[: nb * 16].reshape(nb, 16), b[:, 6:22])
        assert np.array_equal(planar[nb * 16: nb * 20].reshape(nb, 4), b[:, 2:6])
        assert np.array_equal(planar[nb * 20:].reshape(nb, 2), b[:, 0:2])
    elif t == ka.Q4_0:
        assert np.array_equal(planar[: nb * 16].reshape(nb, 16), b[:, 2:18])
        assert np.array_equal(planar[nb * 16:].reshape(nb, 2), b[:, 0:2])
    elif t == ka.Q8_0:
        assert np.array_equal(planar[: nb * 32].reshape(nb, 32), b[:, 2:34])
        assert np.array_equal(planar[nb * 32:].reshape(nb, 2), b[:, 0:2])
    else:                 # Q4_K: d,dmin(4) scales(12) qs(128) -> qs | scales | dm
        assert np.array_equal(planar[: nb * 128].reshape(nb, 128), b[:, 16:144])
        assert np.array_equal(planar[nb * 128: nb * 140].reshape(nb, 12), b[:, 4:16])
        assert np.array_equal(planar[nb * 140:].reshape(nb, 4), b[:, 0:4])


def test_repack_rejects_partial_blocks():
    assert ka.lib().mi355x_repack_to_planar(ka.Q5_0, None, None, 33) != 0
    assert ka.lib().mi355x_repack_to_planar(ka.F32, None, None, 32) != 0
    assert ka.lib().mi355x_type_row_bytes(ka.Q5_0, 1280) == 880
    assert ka.lib().mi355x_type_row_bytes(ka.Q4_K, 1280) == 720
    assert ka.lib().mi355x_type_row_bytes(ka.Q4_K, 384) == 0       # not a multiple of 256 (tiny.en cannot be Q4_K)


def test_device_gelu_table_equals_reference_table():
    tab = np.zeros(65536, dtype=np.uint16)
    ka.lib().mi355x_gelu_table_host(ptr(tab))
    ref = np.load(G / "blocks.npz")["gelu_table"]
    x = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    ok = np.isfinite(x)
    assert np.array_equal(tab[ok], ref[ok])


def test_no_gpu_means_no_context():
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    assert ka.lib().mi355x_device_count() == 0
    with pytest.raises(RuntimeError):
        ka.Ctx(0)                                                   # the product path fails loudly, no CPU fallback


def test_synthetic_model_loads_in_reference(tmp_path):
    exe = ROOT / "oracle" / "_ref" / "cpu_baseline"
    if not exe.exists():
        pytest.skip("oracle/_ref not built")
    from whisper_cpp_amd.synth_model import make_model
    m = make_model("micro", "q5_0", out_dir=tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "oracle" / "_ref"))
    r = subprocess.run([str(exe), str(m), "4", "2", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["encode_ms"] > 0 and d["decode_ms_per_token"] > 0 and "AVX2 = 1" in d["system_info"]


def test_algorithmic_figures_match_survey():
    sys.path.insert(0, str(ROOT))
    import bench
    f = bench.algorithmic_figures("large-v3", "q5_0")
    assert abs(f["decode_bytes_per_token"] / 1e6 - 802) < 8          # ~802 MB/token
    assert abs(f["encode_flop"] / 1e12 - 2.597) < 0.03               # ~2.597 TFLOP per encode
    f = bench.algorithmic_figures("large-v3-turbo", "q8_0")
    assert abs(f["decode_bytes_per_token"] / 1e6 - 200) < 4
    f = bench.algorithmic_figures("base.en", "q5_0")
    assert abs(f["decode_bytes_per_token"] / 1e6 - 52.3) < 1.0


def test_stream_assignment():
    from whisper_cpp_amd.dist_timing import aggregate, assign_streams
    assert assign_streams(8, 8) == [[i] for i in range(8)]
    assert assign_streams(8, 2) == [[0, 2, 4, 6], [1, 3, 5, 7]]
    ms, agg, cps = aggregate(2.0, 4, 8)
    assert ms == 500.0 and agg == 62.5 and cps == 16.0


_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import __graft_entry__ as g
g.load_package()
from whisper_cpp_amd.dist_timing import timed_region, aggregate
dist.init_process_group("gloo")
rank = dist.get_rank()
el = timed_region(lambda: time.sleep(0.05 * (rank + 1)), 3, dist)
ms, agg, cps = aggregate(el, 3, dist.get_world_size())
open(os.path.join({out!r}, "r%d.txt" % rank), "w").write("%d %.6f %.3f" % (rank, el, agg))
dist.destroy_process_group()
"""


def test_two_rank_timing_takes_max_over_ranks(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=str(ROOT), out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    files = sorted(tmp_path.glob("r*.txt"))
    assert len(files) == 2, r.stdout[-800:]
    el = [float(f.read_text().split()[1]) for f in files]
    assert abs(el[0] - el[1]) < 1e-9                      # MAX-reduced: identical on both ranks
    assert 0.29 <= el[0] < 0.6                            # the slow rank (3 x 0.1 s) sets the time
