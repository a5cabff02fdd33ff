"""GPU tests of the int8 tile GEMM over the quantized operands (csrc/kernels/mmq.hip; run with `-m gpu` on an MI355X), through the
C ABI of include/mi355x_kernels.h.

  * activation ROWS (mi355x_prep_act modes 3 / 4): bit-exact against the oracle's quantize_row_q8_0 / quantize_row_q8_K
    (arch/x86/quants.c:302-398, ggml-quants.c:2768-2805) — integers, scales and block sums;
  * mi355x_gemm_q8act against oracle_mul_mat (the CPU's integer block dots with f32 scale-accumulate, ggml-cpu/ggml-cpu.c:1322-1357)
    at NMSE <= 1e-10 — the bar of the T <= 8 mat-vec kernels (tests/test_gpu.py::test_mul_mat_vs_oracle), not the 2e-6 of the f16
    MFMA path — for T in {9, 64, 257, 1500} on all four formats; the tile shapes and the two scale forms (rank-1 MFMA / VALU) are
    bit-identical to each other (same integers, same f32 order);
  * epilogues (bias / scale / GELU / residual / F16 destination / per-column bias) against the plain product;
  * producers that leave the rows in passing — LayerNorm, the attention kernel, the GEMM's own epilogue (fc1 -> fc2) — are
    bit-identical to a separate mi355x_prep_act pass over their F32 result.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import nmse, ptr

pytestmark = pytest.mark.gpu

QT = {"q4_0": 2, "q5_0": 6, "q8_0": 8, "q4_K": 12}


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from whisper_cpp_amd import kernels_api as ka
    ctx = ka.Ctx(0)
    yield ctx, ka, torch
    ctx.close()


def dev(torch, a):
    t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    torch.cuda.synchronize()
    return t


def quantize(oracle, ka, tid, wf):
    """ggml blocks + planar bytes of a weight matrix (tests/test_gpu.py::quantize: Q4_K super-blocks are drawn directly, any bit pattern)"""
    N, K = wf.shape
    if tid == 12:
        rng = np.random.default_rng(N * 31 + K)
        nblk = N * K // 256
        blk = np.zeros((nblk, 144), dtype=np.uint8)
        d = (rng.uniform(0.5, 1.5, nblk) / (40.0 * np.sqrt(K) * 8)).astype(np.float16)
        dmin = (rng.uniform(0.5, 1.5, nblk) / (40.0 * np.sqrt(K))).astype(np.float16)
        blk[:, 0:2] = d.view(np.uint8).reshape(nblk, 2)
        blk[:, 2:4] = dmin.view(np.uint8).reshape(nblk, 2)
        blk[:, 4:144] = rng.integers(0, 256, (nblk, 140), dtype=np.uint8)
        blocks = blk.ravel()
    else:
        blocks = np.empty(N * ka.row_bytes(tid, K), dtype=np.uint8)
        oracle.oracle_quantize_row_ref(tid, ptr(wf), ptr(blocks), N * K)
    return blocks, ka.repack_to_planar(tid, blocks, N * K)


def rows_views(buf: np.ndarray, q8k: bool, K: int, T: int):
    """(q int8 [T][K], d f32 [T][nd], bsum i32 [T][K/32] or None) of a rows buffer (csrc/kernels/qrows.h: the scales and sums are stored
    block-major, [nd][T] / [K/32][T]; returned per column here)"""
    nd = K // 256 if q8k else K // 32
    q = buf[: T * K].view(np.int8).reshape(T, K)
    d = np.ascontiguousarray(buf[T * K: T * K + T * nd * 4].view(np.float32).reshape(nd, T).T)
    bs = np.ascontiguousarray(buf[T * K + T * nd * 4: T * K + T * nd * 4 + T * (K // 32) * 4].view(np.int32).reshape(K // 32, T).T) if q8k else None
    return q, d, bs


def make_rows(gpu, tid, x_d, K, T):
    ctx, ka, torch = gpu
    nbytes = ka.lib().mi355x_act_rows_bytes(tid, K, T)
    r_d = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_prep_act(ctx.h, x_d.data_ptr(), K * 4, 0, r_d.data_ptr(), K, T, 4 if tid == 12 else 3), "prep_act rows")
    return r_d


def run_mmq(gpu, tid, planar, x, K, N, T, ep=None, dst_f16=False, prep=False, prep_only=False):
    ctx, ka, torch = gpu
    w_d, x_d = dev(torch, planar), dev(torch, x)
    r_d = make_rows(gpu, tid, x_d, K, T)
    y_d = torch.full((T, N), 3.0, dtype=torch.float16 if dst_f16 else torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    tw = ka.tensor(w_d.data_ptr(), tid, [K, N])
    p_d = None
    if prep:
        p_d = torch.zeros(ka.lib().mi355x_act_rows_bytes(ka.Q5_0, N, T), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ctx.check(ka.lib().mi355x_gemm_q8act_prep(ctx.h, C.byref(tw), r_d.data_ptr(), T, None if prep_only else y_d.data_ptr(), N * 4, ep, p_d.data_ptr()), "gemm_q8act_prep")
    else:
        ctx.check(ka.lib().mi355x_gemm_q8act(ctx.h, C.byref(tw), r_d.data_ptr(), T, y_d.data_ptr(), N * (2 if dst_f16 else 4), ka.F16 if dst_f16 else ka.F32, ep), "gemm_q8act")
    ctx.sync()
    return y_d.cpu().numpy(), (p_d.cpu().numpy() if p_d is not None else None)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,T", [(1280, 9), (512, 300), (5120, 64), (256, 1500)])
def test_activation_rows_are_the_oracles_blocks_bit_for_bit(gpu, oracle, K, T):
    ctx, ka, torch = gpu
    rng = np.random.default_rng(K + T)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.01, 30.0, (T, 1))).astype(np.float32)
    x[0, :32] = 0.0                                     # an all-zero Q8_0 block (amax == 0)
    if T > 2:
        x[2, :] = 0.0                                   # an all-zero Q8_K super-block
    x[1, 5] = 1e4                                       # one dominating element: everything else in its block rounds to 0
    x_d = dev(torch, x)
    # Q8_0 rows
    r = make_rows(gpu, ka.Q5_0, x_d, K, T)
    ctx.sync()
    q, d, _ = rows_views(r.cpu().numpy(), False, K, T)
    blk = np.empty(T * K // 32 * 34, dtype=np.uint8)
    oracle.oracle_quantize_row_q8_0(ptr(x), ptr(blk), T * K)
    b = blk.reshape(T, K // 32, 34)
    assert np.array_equal(q, b[:, :, 2:].view(np.int8).reshape(T, K))
    assert np.array_equal(d.view(np.uint32), b[:, :, :2].copy().view(np.float16).reshape(T, K // 32).astype(np.float32).view(np.uint32))
    # Q8_K rows
    r = make_rows(gpu, ka.Q4_K, x_d, K, T)
    ctx.sync()
    q, d, bs = rows_views(r.cpu().numpy(), True, K, T)
    blk = np.empty(T * K // 256 * 292, dtype=np.uint8)          # block_q8_K: f32 d, int8 qs[256], int16 bsums[16] (ggml-common.h:371-376)
    oracle.oracle_quantize_row_q8_K(ptr(x), ptr(blk), T * K)
    b = blk.reshape(T, K // 256, 292)
    assert np.array_equal(q, b[:, :, 4:260].view(np.int8).reshape(T, K))
    assert np.array_equal(d.view(np.uint32), b[:, :, :4].copy().view(np.float32).reshape(T, K // 256).view(np.uint32))
    bsum16 = b[:, :, 260:292].copy().view(np.int16).reshape(T, K // 256, 8, 2).astype(np.int32).sum(axis=-1).reshape(T, K // 32)
    assert np.array_equal(bs, bsum16)


@pytest.mark.parametrize("t", list(QT))
@pytest.mark.parametrize("K,N,T", [(1280, 640, 9), (512, 515, 64), (1280, 384, 257), (5120, 130, 64), (1280, 1280, 1500), (256, 33, 40)])
def test_mmq_vs_oracle(gpu, oracle, t, K, N, T):
    """the int8 tile GEMM gives the CPU's integer block sums: the T <= 8 bar (1e-10), not the f16 MFMA path's 2e-6"""
    _, ka, _ = gpu
    tid = QT[t]
    rng = np.random.default_rng(K * 7 + N * 3 + T)
    wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[0, :32] = 0.0
    blocks, planar = quantize(oracle, ka, tid, wf)
    ref = np.empty((T, N), dtype=np.float32)
    oracle.oracle_mul_mat(tid, ptr(blocks), ptr(x), ptr(ref), K, N, T)
    got, _ = run_mmq(gpu, tid, planar, x, K, N, T)
    assert np.isfinite(got).all()
    e = nmse(ref, got)
    assert e < 1e-10, f"{t} K={K} N={N} T={T}: NMSE {e:.3e}"
    # per column too: a wrong column must not hide behind 1499 right ones
    worst = max(nmse(ref[c], got[c]) for c in range(0, T, max(1, T // 37)))
    assert worst < 1e-9, f"{t} K={K} N={N} T={T}: worst sampled column NMSE {worst:.3e}"


@pytest.mark.parametrize("t", ["q4_0", "q5_0", "q8_0"])
@pytest.mark.parametrize("K,N,T", [(1280, 384, 300), (512, 200, 77), (5120, 256, 130)])
def test_mmq_tile_shapes_and_scale_forms_are_bit_identical(gpu, oracle, t, K, N, T):
    _, ka, _ = gpu
    tid = QT[t]
    rng = np.random.default_rng(K + N + T + tid)
    wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((T, K)).astype(np.float32)
    _, planar = quantize(oracle, ka, tid, wf)
    got = {}
    # (kernel library test options, include/mi355x_kernels.h: 0 MMQ_GROUP, 1 MMQ_SCALE_MFMA, 2 MMQ_TILE)
    for name, (which, value) in {"default": (None, 0), "128x64": (2, 12864), "valu-scales": (1, 0), "single launches": (0, 0)}.items():
        if which is not None:
            ka.lib().mi355x_test_option(which, value, 1)
        try:
            got[name], _ = run_mmq(gpu, tid, planar, x, K, N, T)
        finally:
            if which is not None:
                ka.lib().mi355x_test_option(which, 0, 0)
    for name in got:
        assert np.array_equal(got["default"].view(np.uint32), got[name].view(np.uint32)), name


@pytest.mark.parametrize("t", ["q5_0", "q8_0", "q4_K"])
def test_grouped_mmq_launch_is_bit_identical_to_single_launches(gpu, oracle, t):
    """three products on the same activation rows (an encoder layer's Q / K / V) leave as ONE grouped launch; every tile runs the single
    form's code: the results are word for word those of three launches"""
    ctx, ka, torch = gpu
    tid = QT[t]
    K, N, T = 512, 384, 333
    rng = np.random.default_rng(11 + tid)
    x = rng.standard_normal((T, K)).astype(np.float32)
    x_d = dev(torch, x)
    r_d = make_rows(gpu, tid, x_d, K, T)
    ws, biases = [], []
    for i in range(3):
        wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        ws.append(dev(torch, quantize(oracle, ka, tid, wf)[1]))
        biases.append(dev(torch, rng.standard_normal(N).astype(np.float32)))
    out = {}
    for grouped in (1, 0):
        ys = [torch.full((T, N), 5.0, dtype=torch.float16 if i == 1 else torch.float32, device="cuda:0") for i in range(3)]
        torch.cuda.synchronize()
        ka.lib().mi355x_test_option(0, grouped, 1)            # MI355X_OPT_MMQ_GROUP
        try:
            ctx.prof(True); ctx.prof_reset()
            for i in range(3):
                ep = ka.Epilogue(bias=biases[i].data_ptr())
                tw = ka.tensor(ws[i].data_ptr(), tid, [K, N])
                ctx.check(ka.lib().mi355x_gemm_q8act(ctx.h, C.byref(tw), r_d.data_ptr(), T, ys[i].data_ptr(), N * (2 if i == 1 else 4), ka.F16 if i == 1 else ka.F32, C.byref(ep)), "gemm_q8act")
            ctx.sync()
            rows = ctx.prof_report(); ctx.prof(False)
        finally:
            ka.lib().mi355x_test_option(0, 0, 0)
        launches = sum(r["calls"] for r in rows if "mmq" in r["name"])
        assert launches == (1 if grouped else 3), rows
        out[grouped] = [y.cpu().numpy() for y in ys]
    for i in range(3):
        assert np.array_equal(out[1][i].view(np.uint16 if i == 1 else np.uint32), out[0][i].view(np.uint16 if i == 1 else np.uint32)), i


def gelu_table_lookup(ka, x: np.ndarray) -> np.ndarray:
    tab = np.empty(65536, dtype=np.uint16)
    ka.lib().mi355x_gelu_table_host(tab.ctypes.data_as(C.c_void_p))
    h = x.astype(np.float16).view(np.uint16)
    y = tab[h].view(np.float16).astype(np.float32)
    y = np.where(x <= -10.0, 0.0, np.where(x >= 10.0, x, y)).astype(np.float32)
    return y


@pytest.mark.parametrize("t", list(QT))
def test_mmq_epilogues_equal_the_separate_ops(gpu, oracle, t):
    """dst = gelu((W x + bias) * scale) + residual, each step rounded to f32 like the separate ggml nodes; F16 destination; per-column bias"""
    ctx, ka, torch = gpu
    tid = QT[t]
    K, N, T = 512, 260, 150
    rng = np.random.default_rng(99 + tid)
    wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((T, K)).astype(np.float32)
    _, planar = quantize(oracle, ka, tid, wf)
    plain, _ = run_mmq(gpu, tid, planar, x, K, N, T)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((T, N)).astype(np.float32)
    b_d, r_d = dev(torch, bias), dev(torch, res)
    ep = ka.Epilogue(bias=b_d.data_ptr(), scale=0.7, has_scale=1, gelu=1, residual=r_d.data_ptr(), residual_nb1=N * 4)
    got, _ = run_mmq(gpu, tid, planar, x, K, N, T, ep=C.byref(ep))
    want = gelu_table_lookup(ka, ((plain + bias[None, :]) * np.float32(0.7)).astype(np.float32)) + res
    assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32))
    # F16 destination (the K / V projections whose ggml_cpy into an F16 cache is folded in) = rounded F32 result
    ep2 = ka.Epilogue(bias=b_d.data_ptr())
    got16, _ = run_mmq(gpu, tid, planar, x, K, N, T, ep=C.byref(ep2), dst_f16=True)
    assert np.array_equal(got16.view(np.uint16), (plain + bias[None, :]).astype(np.float32).astype(np.float16).view(np.uint16))
    # bias per COLUMN (the conv bias form)
    bt = rng.standard_normal(T).astype(np.float32)
    bt_d = dev(torch, bt)
    ep3 = ka.Epilogue(bias=bt_d.data_ptr(), bias_per_col=1)
    got3, _ = run_mmq(gpu, tid, planar, x, K, N, T, ep=C.byref(ep3))
    assert np.array_equal(got3.view(np.uint32), (plain + bt[:, None]).astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("t", ["q5_0", "q8_0", "q4_K"])
@pytest.mark.parametrize("only", [False, True])
def test_mmq_epilogue_leaves_the_next_gemms_rows(gpu, oracle, t, only):
    """fc1 + bias + GELU whose only reader is fc2: the rows written by the epilogue are those of a separate pass over the F32 result"""
    ctx, ka, torch = gpu
    tid = QT[t]
    K, N, T = 256, 512, 200
    rng = np.random.default_rng(5 + tid)
    wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((T, K)).astype(np.float32)
    _, planar = quantize(oracle, ka, tid, wf)
    bias = rng.standard_normal(N).astype(np.float32)
    b_d = dev(torch, bias)
    ep = ka.Epilogue(bias=b_d.data_ptr(), gelu=1)
    full, _ = run_mmq(gpu, tid, planar, x, K, N, T, ep=C.byref(ep))
    got, rows = run_mmq(gpu, tid, planar, x, K, N, T, ep=C.byref(ep), prep=True, prep_only=only)
    if not only:
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
    f_d = dev(torch, full)
    want = make_rows(gpu, ka.Q5_0, f_d, N, T)
    ctx.sync()
    assert np.array_equal(rows, want.cpu().numpy())


@pytest.mark.parametrize("mode,K", [(3, 1280), (4, 1280), (3, 384), (4, 512)])
def test_layernorm_leaves_the_rows_of_its_result(gpu, mode, K):
    ctx, ka, torch = gpu
    T = 333
    rng = np.random.default_rng(K + mode)
    x = (rng.standard_normal((T, K)) * 3 + 0.5).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    b = (0.1 * rng.standard_normal(K)).astype(np.float32)
    x_d, w_d, b_d = dev(torch, x), dev(torch, w), dev(torch, b)
    y_d = torch.zeros((T, K), dtype=torch.float32, device="cuda:0")
    tid = ka.Q4_K if mode == 4 else ka.Q5_0
    nbytes = ka.lib().mi355x_act_rows_bytes(tid, K, T)
    r_d = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    tx, ty = ka.tensor(x_d.data_ptr(), ka.F32, [K, T]), ka.tensor(y_d.data_ptr(), ka.F32, [K, T])
    ctx.check(ka.lib().mi355x_norm_prep(ctx.h, C.byref(tx), C.byref(ty), 1e-5, w_d.data_ptr(), b_d.data_ptr(), r_d.data_ptr(), mode), "norm_prep rows")
    ctx.sync()
    y2_d = torch.zeros((T, K), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_norm(ctx.h, C.byref(tx), C.byref(ka.tensor(y2_d.data_ptr(), ka.F32, [K, T])), 1e-5, w_d.data_ptr(), b_d.data_ptr()), "norm")
    ctx.sync()
    assert np.array_equal(y_d.cpu().numpy().view(np.uint32), y2_d.cpu().numpy().view(np.uint32))
    want = make_rows(gpu, tid, y_d, K, T)
    ctx.sync()
    assert np.array_equal(r_d.cpu().numpy(), want.cpu().numpy())


@pytest.mark.parametrize("T,n_kv,H", [(1500, 1536, 20), (200, 320, 4), (130, 64, 6)])
def test_attention_leaves_the_rows_of_its_result(gpu, T, n_kv, H):
    ctx, ka, torch = gpu
    D = 64
    rng = np.random.default_rng(T + n_kv + H)
    q = (rng.standard_normal((T, H, D)) * 0.6).astype(np.float32)
    k = (rng.standard_normal((n_kv, H, D)) * 0.6).astype(np.float16)
    v = rng.standard_normal((n_kv, H, D)).astype(np.float16)
    q_d, k_d, v_d = dev(torch, q), dev(torch, k), dev(torch, v)
    tq = ka.tensor(q_d.data_ptr(), ka.F32, [D, T, H], [4, H * D * 4, D * 4, T * H * D * 4])
    tk = ka.tensor(k_d.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
    tv = ka.tensor(v_d.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
    o_d = torch.zeros((T, H, D), dtype=torch.float32, device="cuda:0")
    o2_d = torch.zeros((T, H, D), dtype=torch.float32, device="cuda:0")
    r_d = torch.zeros(ka.lib().mi355x_act_rows_bytes(ka.Q5_0, H * D, T), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_flash_attn_ext_prep_rows(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), None, C.byref(ka.tensor(o_d.data_ptr(), ka.F32, [D, H, T])), 0.125, r_d.data_ptr()), "flash_attn rows")
    ctx.check(ka.lib().mi355x_flash_attn_ext(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), None, C.byref(ka.tensor(o2_d.data_ptr(), ka.F32, [D, H, T])), 0.125), "flash_attn")
    ctx.sync()
    assert np.array_equal(o_d.cpu().numpy().view(np.uint32), o2_d.cpu().numpy().view(np.uint32))
    want = make_rows(gpu, ka.Q5_0, o_d.reshape(T, H * D), H * D, T)
    ctx.sync()
    assert np.array_equal(r_d.cpu().numpy(), want.cpu().numpy())


def test_mmq_full_size_encoder_products(gpu, oracle):
    """large-v3's fc1 (1280 -> 5120, 1500 columns) against the f64 product of the dequantized operands: the kernel's error is the f32
    accumulation's (NMSE ~1e-13), far below the 2e-6 the f16 path is held to"""
    ctx, ka, torch = gpu
    K, N, T = 1280, 5120, 1500
    rng = np.random.default_rng(2024)
    wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((T, K)).astype(np.float32)
    blocks, planar = quantize(oracle, ka, ka.Q5_0, wf)
    got, _ = run_mmq(gpu, ka.Q5_0, planar, x, K, N, T)
    wq = np.empty((N, K), dtype=np.float32)
    oracle.oracle_dequantize_row(ka.Q5_0, ptr(blocks), ptr(wq), N * K)
    xb = np.empty(T * K // 32 * 34, dtype=np.uint8)
    oracle.oracle_quantize_row_q8_0(ptr(x), ptr(xb), T * K)
    xq = np.empty((T, K), dtype=np.float32)
    oracle.oracle_dequantize_row(ka.Q8_0, ptr(xb), ptr(xq), T * K)
    want = xq.astype(np.float64) @ wq.astype(np.float64).T
    assert nmse(want, got) < 1e-11
