"""Pin the CPU oracle (oracle/oracle.c) against vectors generated from the reference itself.

tests/golden/blocks.npz and ops.npz were produced by tests/golden/make_golden.py from oracle/_ref, i.e. by the
reference's own compiled kernels (AVX2 build).  The reference has no known-answer vectors for this path
(SURVEY.md §8c), so "the reference run here" is the pin.
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from conftest import nmse, ptr

G = Path(__file__).resolve().parent / "golden"
TYPES = {"q4_0": 2, "q5_0": 6, "q8_0": 8, "q4_K": 12}


@pytest.fixture(scope="module")
def blocks():
    return np.load(G / "blocks.npz")


@pytest.fixture(scope="module")
def ops():
    z = np.load(G / "ops.npz")
    man = json.loads(bytes(z["manifest"]).decode())
    return z, {m["case"]: m for m in man}


def test_f16_conversion_matches_reference(oracle, blocks):
    x = blocks["x"]
    got = np.array([oracle.oracle_f32_to_f16(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got, blocks["x_f16"])
    # exhaustive round trip: every finite half survives f16 -> f32 -> f16, and matches numpy's conversion
    allh = np.arange(65536, dtype=np.uint16)
    f = np.array([oracle.oracle_f16_to_f32(int(h)) for h in allh], dtype=np.float32)
    ref = allh.view(np.float16).astype(np.float32)
    fin = np.isfinite(ref)
    assert np.array_equal(f[fin], ref[fin])
    # f32 -> f16 on a dense sample incl. ties, subnormals and overflow
    rng = np.random.default_rng(1)
    s = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 10 ** rng.uniform(-9, 6, 20000).astype(np.float32),
                        np.array([65504, 65519.99, 65520, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 0.0, -0.0, 6.1e-5], dtype=np.float32)])
    got = np.array([oracle.oracle_f32_to_f16(float(v)) for v in s], dtype=np.uint16)
    with np.errstate(over="ignore"):
        assert np.array_equal(got, s.astype(np.float16).view(np.uint16))


@pytest.mark.parametrize("t", list(TYPES))
def test_dequantize_bit_exact(oracle, blocks, t):
    blk = blocks[f"wblk_{t}"]
    n = blocks["w"].size
    y = np.zeros(n, dtype=np.float32)
    oracle.oracle_dequantize_row(TYPES[t], ptr(blk), ptr(y), n)
    ref = blocks[f"wdeq_{t}"]
    if t == "q4_K":
        # the reference is compiled with -ffp-contract=fast: d1*q - m1 may be a single fma there (1 ulp)
        assert np.allclose(y, ref, rtol=0, atol=np.abs(ref).max() * 2.0 ** -22)
    else:
        assert np.array_equal(y, ref)


@pytest.mark.parametrize("t", ["q4_0", "q5_0", "q8_0", "q4_K"])
def test_weight_quantizer_bit_exact(oracle, blocks, t):
    w = blocks["w"]
    blk = np.zeros_like(blocks[f"wblk_{t}"])
    oracle.oracle_quantize_row_ref(TYPES[t], ptr(w), ptr(blk), w.size)
    assert np.array_equal(blk, blocks[f"wblk_{t}"])


def test_q4_K_weight_quantizer_search_bit_exact(oracle, blocks):
    """quantize_row_q4_K_ref incl. the make_qkx2_quants grid search (ggml-quants.c:799-878, :1457-1527) on 16 super-blocks
    with per-block magnitudes over 3 decades, an all-zero super-block, an all-positive block (min clamps to 0), a constant
    block and an outlier block — against the bytes the reference itself produced (tests/golden/make_golden.py)"""
    x = blocks["q4k_x"]
    blk = np.zeros_like(blocks["q4k_blk"])
    oracle.oracle_quantize_row_ref(TYPES["q4_K"], ptr(x), ptr(blk), x.size)
    assert np.array_equal(blk, blocks["q4k_blk"])
    # and the round trip stays within the format's resolution per super-block: half a 4-bit step of the widest 32-block,
    # plus the 6-bit rounding of the scale (x15 levels) and of the min, with 50 % slack for the fitted (not min/max) grids
    y = np.zeros_like(x)
    oracle.oracle_dequantize_row(TYPES["q4_K"], ptr(blk), ptr(y), x.size)
    err = np.abs(x - y).reshape(-1, 256).max(axis=1)
    xb = x.reshape(-1, 8, 32)
    lo = np.minimum(xb.min(axis=2), 0)
    span = (xb.max(axis=2) - lo).max(axis=1)
    negmin = (-lo).max(axis=1)
    bound = 1.5 * (span / 30 + span / 126 + negmin / 126) + 1e-6
    assert np.all(err <= bound), float((err / bound).max())


def test_activation_quantizers_bit_exact(oracle, blocks):
    x = blocks["x"]
    a8 = np.zeros_like(blocks["act_q8_0"])
    oracle.oracle_quantize_row_q8_0(ptr(x), ptr(a8), x.size)
    assert np.array_equal(a8, blocks["act_q8_0"]), "Q8_0 activation blocks differ from the AVX2 reference"
    aK = np.zeros_like(blocks["act_q8_K"])
    oracle.oracle_quantize_row_q8_K(ptr(x), ptr(aK), x.size)
    assert np.array_equal(aK, blocks["act_q8_K"])


@pytest.mark.parametrize("t", list(TYPES))
def test_vec_dot(oracle, blocks, t):
    act = blocks["act_q8_K"] if t == "q4_K" else blocks["act_q8_0"]
    got = oracle.oracle_vec_dot(TYPES[t], blocks["w"].size, ptr(blocks[f"wblk_{t}"]), ptr(act))
    ref = float(blocks[f"dot_{t}"])
    # identical integer sums; only the f32 accumulation order differs (8 SIMD lanes vs sequential)
    assert abs(got - ref) <= 2e-6 * max(1.0, abs(ref)), (got, ref)


def test_gelu_table_bit_exact(oracle, blocks):
    allh = np.arange(65536, dtype=np.uint16)
    x = allh.view(np.float16).astype(np.float32)
    ok = np.isfinite(x) & (np.abs(x) < 10)
    xs = np.ascontiguousarray(x[ok])
    y = np.zeros_like(xs)
    oracle.oracle_gelu(ptr(xs), ptr(y), xs.size)
    ref = blocks["gelu_table"][ok].view(np.float16).astype(np.float32)
    assert np.array_equal(y, ref)


def _leaf(z, man, case, i, dtype=np.uint8):
    return np.ascontiguousarray(z[f"{case}.leaf{i}"]).view(dtype)


@pytest.mark.parametrize("t,tid", [("q5_0", 6), ("q8_0", 8), ("q4_0", 2), ("q4_K", 12), ("f16", 1)])
def test_mul_mat_matches_reference_graph(oracle, ops, t, tid):
    z, man = ops
    case = f"golden_mul_mat_{t}"
    m = man[case]
    # the two leaves are created inside one C++ expression (unspecified order): identify them by type
    iw = [i for i, l in enumerate(m["leaves"]) if l["type"] == tid][0]
    ix = 1 - iw
    K, N = m["leaves"][iw]["ne"][:2]
    T = m["leaves"][ix]["ne"][1]
    w = _leaf(z, man, case, iw)
    x = _leaf(z, man, case, ix, np.float32)
    ref = z[f"{case}.out0"]
    got = np.zeros(N * T, dtype=np.float32)
    oracle.oracle_mul_mat(tid, ptr(w), ptr(x), ptr(got), K, N, T)
    assert nmse(ref, got) < 1e-12, nmse(ref, got)
    assert np.abs(ref - got).max() <= 3e-6 * np.abs(ref).max()


def test_norm_gelu_softmax_im2col(oracle, ops):
    z, man = ops
    x = _leaf(z, man, "golden_norm", 0, np.float32)
    y = np.zeros_like(x)
    oracle.oracle_norm(ptr(x), ptr(y), 384, 5, 1e-5)
    assert nmse(z["golden_norm.out0"], y) < 1e-12
    x = _leaf(z, man, "golden_gelu", 0, np.float32)
    y = np.zeros_like(x)
    oracle.oracle_gelu(ptr(x), ptr(y), x.size)
    assert np.array_equal(y, z["golden_gelu.out0"])
    x = _leaf(z, man, "golden_soft_max", 0, np.float32)
    mk = _leaf(z, man, "golden_soft_max", 1, np.float32)
    y = np.zeros_like(x)
    oracle.oracle_soft_max(ptr(x), ptr(mk), ptr(y), 100, 6, 0.3)
    assert nmse(z["golden_soft_max.out0"], y) < 1e-12
    x = _leaf(z, man, "golden_im2col", 1, np.float32)
    dst = np.zeros(30 * 24, dtype=np.uint16)          # OW = (50 + 2*1 - 3)/2 + 1 = 25 ... computed below
    OW = man["golden_im2col"]["outs"][0]["ne"][1]
    dst = np.zeros(OW * 30, dtype=np.uint16)
    oracle.oracle_im2col_1d_f16(ptr(x), ptr(dst), 50, 10, OW, 3, 2, 1, 1)
    assert np.array_equal(dst.view(np.float16).astype(np.float32), z["golden_im2col.out0"])


@pytest.mark.parametrize("mode", [0, 2])
def test_rope(oracle, ops, mode):
    z, man = ops
    case = f"golden_rope_mode{mode}"
    x = _leaf(z, man, case, 0, np.float32)
    pos = _leaf(z, man, case, 1, np.int32)
    y = np.zeros_like(x)
    oracle.oracle_rope(ptr(x), ptr(pos), ptr(y), 64, 3, 7, 48, mode, 4096, 10000.0, 0.5, 1.0, 1.0, 32.0, 1.0)
    assert nmse(z[f"{case}.out0"], y) < 1e-10


def test_flash_attn(oracle, ops):
    z, man = ops
    case = "golden_flash_attn"
    D, T, H, n_kv = 64, 3, 2, 40
    q = _leaf(z, man, case, 0, np.float32)       # [T][H][D]
    k = _leaf(z, man, case, 1, np.uint16)        # [n_kv][H][D]
    v = _leaf(z, man, case, 2, np.uint16)
    mf = _leaf(z, man, case, 3, np.float32)      # [T][n_kv]
    mh = mf.astype(np.float16).view(np.uint16)
    out = np.zeros(T * H * D, dtype=np.float32)
    oracle.oracle_flash_attn(ptr(q), ptr(k), ptr(v), ptr(mh), ptr(out), D, T, H, n_kv, 0.125)
    # same algorithm incl. f16 V accumulation; the reference's f16 dot / mad run in 8-lane SIMD order
    assert nmse(z[f"{case}.out0"], out) < 1e-5


FA_CASES = {   # case: (T, H, n_kv, has_mask) — the three arithmetic paths of the CPU dispatcher (ggml-cpu/ops.cpp:9077-9230), 8 threads
    "golden_flash_attn": (3, 2, 40, True), "golden_fattn_split": (1, 3, 1536, False), "golden_fattn_split_masked": (1, 2, 600, True),
    "golden_fattn_vec5": (5, 2, 1536, False), "golden_fattn_tiled": (70, 2, 200, True), "golden_fattn_tiled_nomask": (64, 2, 136, False),
}


@pytest.mark.parametrize("case", list(FA_CASES))
def test_flash_attn_dispatcher_is_bit_exact(oracle, ops, case):
    """oracle_flash_attn_ext restates the reference's flash_attn_ext PER PATH (split-KV over the thread count for T == 1 and
    n_kv >= 512, the F32 tiled path for T >= 64, the F16-accumulating vec path otherwise) in the AVX2 lane order:
    every output word equals the reference's (vectors generated from oracle/_ref by tests/golden/make_golden.py)"""
    z, man = ops
    T, H, n_kv, has_mask = FA_CASES[case]
    D = 64
    q = _leaf(z, man, case, 0, np.float32)
    k = _leaf(z, man, case, 1, np.uint16)
    v = _leaf(z, man, case, 2, np.uint16)
    mh = _leaf(z, man, case, 3, np.float32).astype(np.float16).view(np.uint16).copy() if has_mask else None
    out = np.zeros(T * H * D, dtype=np.float32)
    oracle.oracle_flash_attn_ext(ptr(q), ptr(k), ptr(v), ptr(mh) if has_mask else None, ptr(out), D, T, H, n_kv, 0.125, 8)
    ref = z[f"{case}.out0"]
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    if case == "golden_fattn_split":
        # the split-KV result depends on the thread count (chunk = ceil(n_kv / nth)): 4 threads give different words
        out4 = np.zeros_like(out)
        oracle.oracle_flash_attn_ext(ptr(q), ptr(k), ptr(v), None, ptr(out4), D, T, H, n_kv, 0.125, 4)
        assert not np.array_equal(out4.view(np.uint32), ref.view(np.uint32)) and nmse(ref, out4) < 1e-4


def test_v_expf_lane(oracle):
    """one lane of ggml_v_expf (vec.h:1215-1252): max error 1.45 + 0.5 ulp by its own comment; exact at 0, flushes like the reference"""
    assert oracle.oracle_v_expf(0.0) == 1.0
    assert oracle.oracle_v_expf(float("-inf")) == 0.0
    assert oracle.oracle_v_expf(-200.0) == 0.0
    xs = np.linspace(-87.0, 0.0, 4001, dtype=np.float32)
    got = np.array([oracle.oracle_v_expf(float(x)) for x in xs], dtype=np.float32)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 3 * 2.0 ** -24


def _mel_golden():
    z = np.load(G / "mel.npz")
    n_len = int(z["n_len"])
    mel = np.full((z["filters"].shape[0], n_len), z["tail_value"], dtype=np.float32)
    mel[:, :z["mel_head"].shape[1]] = z["mel_head"]
    return z, mel


def test_log_mel_matches_the_reference_front_end_on_real_speech(oracle):
    """oracle_log_mel (recursive FFT and all) against whisper's own log_mel_spectrogram run on samples/jfk.wav with the real 80-band
    filterbank (golden generated through oracle/_ref/mel_ref).  The reference build contracts a*b + c*d chains of its butterflies
    into fmas (-ffp-contract=fast), the oracle is compiled without contraction: agreement to a few f32 roundings of the spectrum."""
    z, want = _mel_golden()
    pcm = (z["pcm16"].astype(np.float32) / 32768.0)
    filt = np.ascontiguousarray(z["filters"])
    n_len = oracle.oracle_log_mel_n_len(len(pcm))
    assert n_len == want.shape[1] == 4100 and int(z["n_len_org"]) == 1099
    out = np.zeros((filt.shape[0], n_len), dtype=np.float32)
    oracle.oracle_log_mel(ptr(pcm), len(pcm), ptr(filt), filt.shape[0], ptr(out))
    assert np.abs(out - want).max() < 2e-5, np.abs(out - want).max()
    assert nmse(want, out) < 1e-12
    assert np.array_equal(out[:, 1104:], want[:, 1104:])        # frames no sample reaches: the clamped floor, exactly
