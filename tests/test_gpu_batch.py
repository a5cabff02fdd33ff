"""GPU tests of the T = 3..8 / cross-state decoder pipeline (whisper.cpp_amd/csrc/kernels/decode_q.hip) through the kernel C ABI.

The claim under test is stronger than a tolerance: a column's result is BIT-IDENTICAL whether it is computed alone by the fused
T = 1 kernels (decode.hip: LayerNorm / attention combine + Q8 quantization in every workgroup) or as one of T columns of the
pre-quantized-activation pipeline (mi355x_act_prepare -> planes -> mi355x_gemv_fused(x_planes)), at any column position and with
per-column destination pointers.  That is what makes a cross-state batch (several whisper_states' decode steps as the columns of
one launch chain) indistinguishable from running every state alone.  The oracle (CPU restatement) pins the values themselves.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import nmse, ptr
from test_gpu import QT, dev, gpu, quantize  # noqa: F401  (gpu is a fixture)

pytestmark = pytest.mark.gpu

# whole-model tests below: ONE chain as wide as the states allow (the default splits them two thirds / one third over two chains; that
# arrangement is covered by tests/test_gpu.py::test_concurrent_streams_on_one_gpu_match_serial).  Read once per process by the plugin.
os.environ.setdefault("GGML_MI355X_BATCH_COLS", "32")


def _planes(ctx, ka, which=0):
    p = ka.lib().mi355x_act_scratch(ctx.h, which)
    assert p, "mi355x_act_scratch failed"
    return p


def _seg(ka, d, s, w_d, tid, N, y_d, dst_type=None, ep=None, nb1=None):
    d.seg[s].w, d.seg[s].wtype, d.seg[s].N = w_d.data_ptr(), tid, N
    if ep is not None:
        d.seg[s].ep = ep
    dst_type = ka.F32 if dst_type is None else dst_type
    d.seg[s].dst, d.seg[s].dst_type = (y_d.data_ptr() if y_d is not None else 0), dst_type
    d.seg[s].dst_nb1 = nb1 if nb1 is not None else N * (4 if dst_type == ka.F32 else 2)


@pytest.mark.parametrize("t", list(QT))
@pytest.mark.parametrize("K,T", [(1280, 1), (1280, 3), (1280, 5), (1280, 8), (512, 4), (512, 7), (1280, 9), (1280, 16), (512, 21), (1280, 32)])      # T > 8: images of 8 columns
def test_layernorm_qkv_through_planes_is_bit_identical_to_the_fused_kernel(gpu, oracle, t, K, T):
    """LN + Q/K/V (three segments, scale / bias epilogues, F16 destinations for K and V = the KV-cache store) — src/whisper.cpp:2529-2598"""
    ctx, ka, torch = gpu
    tid = QT[t]
    rng = np.random.default_rng(K * 7 + T + tid)
    x = (rng.standard_normal((T, K)) * 2).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(K)).astype(np.float32)
    Ns = (K, K, 384)
    ws, biases = [], []
    for N in Ns:
        wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        ws.append(quantize(oracle, ka, tid, wf))
        biases.append((rng.standard_normal(N) * 0.1).astype(np.float32))
    x_d, lw_d, lb_d = dev(torch, x), dev(torch, lw), dev(torch, lb)
    w_d = [dev(torch, p) for _, p in ws]
    b_d = [dev(torch, b) for b in biases]
    types = (ka.F32, ka.F16, ka.F16)

    def desc(Tn):
        d = ka.GemvDesc()
        d.K, d.T, d.nseg = K, Tn, 3
        return d

    def outs(Tn):
        return [torch.zeros((Tn, N), dtype=torch.float32 if ty == ka.F32 else torch.float16, device="cuda:0") for N, ty in zip(Ns, types)]

    def eps(s):
        return ka.Epilogue(b_d[s].data_ptr() if s != 1 else 0, 0.25, 1 if s < 2 else 0, 0, 0, 0)

    # (a) every column alone through the fused kernel
    alone = []
    for c in range(T):
        ys = outs(1)
        d = desc(1)
        d.x, d.x_nb1, d.has_norm, d.eps, d.ln_w, d.ln_b = x_d.data_ptr() + c * K * 4, K * 4, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
        for s in range(3):
            _seg(ka, d, s, w_d[s], tid, Ns[s], ys[s], types[s], eps(s))
        torch.cuda.synchronize()
        ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "gemv_fused")
        ctx.sync()
        alone.append([y.cpu().numpy()[0] for y in ys])
    # (b) all columns through the planes
    planes = _planes(ctx, ka)
    a = ka.ActDesc()
    a.x, a.x_nb1, a.K, a.T, a.wtype, a.has_norm, a.eps, a.ln_w, a.ln_b = x_d.data_ptr(), K * 4, K, T, tid, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
    ctx.check(ka.lib().mi355x_act_prepare(ctx.h, C.byref(a), planes), "act_prepare")
    ys = outs(T)
    d = desc(T)
    d.x_planes = planes
    for s in range(3):
        _seg(ka, d, s, w_d[s], tid, Ns[s], ys[s], types[s], eps(s))
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "gemv_fused(planes)")
    ctx.sync()
    for s in range(3):
        got = ys[s].cpu().numpy()
        for c in range(T):
            assert np.array_equal(got[c].view(np.uint8), alone[c][s].view(np.uint8)), (t, K, T, s, c)
    # (c) the values themselves: oracle norm -> oracle mul_mat (segment 0, before the f32 epilogue is a bias + scale)
    nx = np.empty_like(x)
    oracle.oracle_norm(ptr(x), ptr(nx), K, T, 1e-5)
    nx = (nx * lw[None, :]).astype(np.float32) + lb[None, :]
    ref = np.empty((T, Ns[0]), dtype=np.float32)
    oracle.oracle_mul_mat(tid, ptr(ws[0][0]), ptr(np.ascontiguousarray(nx)), ptr(ref), K, Ns[0], T)
    want = ((ref + biases[0][None, :]).astype(np.float32) * np.float32(0.25)).astype(np.float32)
    assert nmse(want, ys[0].cpu().numpy()) < 1e-9


@pytest.mark.parametrize("t", list(QT))
@pytest.mark.parametrize("K,N,T", [(5120, 1280, 5), (5120, 1280, 8), (1280, 1280, 3), (2048, 512, 6), (1280, 5120, 2), (5120, 1280, 16), (5120, 1280, 11), (1280, 5120, 32), (2048, 512, 24), (5120, 1280, 27)])
def test_plain_mat_vec_through_planes_with_per_column_pointers(gpu, oracle, t, K, N, T):
    """fc2 / any bias + residual projection: columns written to SCATTERED destinations (what a cross-state batch does) equal the fused
    T = 1 result of every column, bit for bit"""
    ctx, ka, torch = gpu
    tid = QT[t]
    rng = np.random.default_rng(K + N + T + tid)
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[1::4] *= np.float32(1e-4)                # columns whose Q8_0 scales are f16 DENORMALS (amax / 127 ~ 2e-6): the matrix-core form (decode_mx.hip) forms dw * dx in an f16 MFMA
    x[2::4, K // 2:] = 0                       # ... and columns with all-zero blocks (scale 0, sums 0)
    wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    blocks, planar = quantize(oracle, ka, tid, wf)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    res = rng.standard_normal((T, N)).astype(np.float32)
    res[1::4] = 0                              # (the tiny columns are compared on their own, not under a residual of order 1)
    x_d, w_d, b_d, r_d = dev(torch, x), dev(torch, planar), dev(torch, bias), dev(torch, res)
    alone = []
    for c in range(T):
        y = torch.zeros((1, N), dtype=torch.float32, device="cuda:0")
        d = ka.GemvDesc()
        d.x, d.x_nb1, d.K, d.T, d.nseg = x_d.data_ptr() + c * K * 4, K * 4, K, 1, 1
        _seg(ka, d, 0, w_d, tid, N, y, ka.F32, ka.Epilogue(b_d.data_ptr(), 0.0, 0, 0, r_d.data_ptr() + c * N * 4, N * 4))
        torch.cuda.synchronize()
        ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "gemv_fused")
        ctx.sync()
        alone.append(y.cpu().numpy()[0])
    planes = _planes(ctx, ka)
    a = ka.ActDesc()
    a.K, a.T, a.wtype = K, T, tid
    for c in range(T):                         # per-column sources, in reverse order of the buffer
        a.xcol[c] = x_d.data_ptr() + (T - 1 - c) * K * 4
    ctx.check(ka.lib().mi355x_act_prepare(ctx.h, C.byref(a), planes), "act_prepare")
    big = torch.zeros((T, 3, N), dtype=torch.float32, device="cuda:0")       # column c lands in big[c, c % 3]
    cols = ka.GemvCols()
    for c in range(T):
        cols.dst[0][c] = big.data_ptr() + ((c * 3) + (c % 3)) * N * 4
        cols.res[0][c] = r_d.data_ptr() + (T - 1 - c) * N * 4
    d = ka.GemvDesc()
    d.K, d.T, d.nseg, d.x_planes, d.cols = K, T, 1, planes, C.addressof(cols)
    _seg(ka, d, 0, w_d, tid, N, None, ka.F32, ka.Epilogue(b_d.data_ptr(), 0.0, 0, 0, 0, 0))
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "gemv_fused(planes, cols)")
    ctx.sync()
    got = big.cpu().numpy()
    for c in range(T):
        assert np.array_equal(got[c, c % 3].view(np.uint32), alone[T - 1 - c].view(np.uint32)), (t, K, N, T, c)
        others = [j for j in range(3) if j != c % 3]
        assert not got[c, others].any()
    ref = np.empty((T, N), dtype=np.float32)
    oracle.oracle_mul_mat(tid, ptr(blocks), ptr(x), ptr(ref), K, N, T)
    want = (ref + bias[None, :]).astype(np.float32) + res
    assert nmse(want, np.stack([alone[c] for c in range(T)])) < 1e-9


@pytest.mark.parametrize("t", ["q4_0", "q5_0", "q8_0"])
@pytest.mark.parametrize("K,N,N2,T,only", [(1280, 5120, 1280, 5, True), (1280, 5120, 1280, 8, False), (512, 2048, 512, 3, True), (384, 1536, 384, 1, False),
                                            (1280, 5120, 1280, 16, True), (1280, 5120, 1280, 13, False), (512, 2048, 512, 32, True)])
def test_producer_epilogue_writes_the_next_mat_vecs_planes(gpu, oracle, t, K, N, N2, T, only):
    """LN + fc1 + bias + GELU whose epilogue leaves the Q8_0 planes of its result; fc2 reads them (src/whisper.cpp:2787-2830).
    Equal, bit for bit, to fused fc1 (F32 result) -> fused fc2 for every column alone."""
    ctx, ka, torch = gpu
    tid = QT[t]
    rng = np.random.default_rng(K + N + T + tid)
    x = (rng.standard_normal((T, K)) * 1.5).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(K)).astype(np.float32)
    w1 = quantize(oracle, ka, tid, (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    w2 = quantize(oracle, ka, tid, (rng.standard_normal((N2, N)) / np.sqrt(N)).astype(np.float32))
    b1 = (rng.standard_normal(N) * 0.1).astype(np.float32)
    b2 = (rng.standard_normal(N2) * 0.1).astype(np.float32)
    x_d, lw_d, lb_d, w1_d, w2_d, b1_d, b2_d = (dev(torch, a) for a in (x, lw, lb, w1[1], w2[1], b1, b2))
    alone_h, alone_y = [], []
    for c in range(T):
        h = torch.zeros((1, N), dtype=torch.float32, device="cuda:0")
        y = torch.zeros((1, N2), dtype=torch.float32, device="cuda:0")
        d = ka.GemvDesc()
        d.x, d.x_nb1, d.K, d.T, d.nseg, d.has_norm, d.eps, d.ln_w, d.ln_b = x_d.data_ptr() + c * K * 4, K * 4, K, 1, 1, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
        _seg(ka, d, 0, w1_d, tid, N, h, ka.F32, ka.Epilogue(b1_d.data_ptr(), 0.0, 0, 1, 0, 0))
        ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "fc1 fused")
        d2 = ka.GemvDesc()
        d2.x, d2.x_nb1, d2.K, d2.T, d2.nseg = h.data_ptr(), N * 4, N, 1, 1
        _seg(ka, d2, 0, w2_d, tid, N2, y, ka.F32, ka.Epilogue(b2_d.data_ptr(), 0.0, 0, 0, x_d.data_ptr() + c * K * 4 if N2 == K else 0, K * 4))
        ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d2)), "fc2 fused")
        ctx.sync()
        alone_h.append(h.cpu().numpy()[0]); alone_y.append(y.cpu().numpy()[0])
    p0, p1 = _planes(ctx, ka, 0), _planes(ctx, ka, 1)
    a = ka.ActDesc()
    a.x, a.x_nb1, a.K, a.T, a.wtype, a.has_norm, a.eps, a.ln_w, a.ln_b = x_d.data_ptr(), K * 4, K, T, tid, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
    ctx.check(ka.lib().mi355x_act_prepare(ctx.h, C.byref(a), p0), "act_prepare")
    h = torch.full((T, N), 7.0, dtype=torch.float32, device="cuda:0")
    y = torch.zeros((T, N2), dtype=torch.float32, device="cuda:0")
    d = ka.GemvDesc()
    d.K, d.T, d.nseg, d.x_planes, d.planes_out, d.planes_out_only = K, T, 1, p0, p1, 1 if only else 0
    _seg(ka, d, 0, w1_d, tid, N, None if only else h, ka.F32, ka.Epilogue(b1_d.data_ptr(), 0.0, 0, 1, 0, 0))
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "fc1 planes -> planes")
    d2 = ka.GemvDesc()
    d2.K, d2.T, d2.nseg, d2.x_planes = N, T, 1, p1
    _seg(ka, d2, 0, w2_d, tid, N2, y, ka.F32, ka.Epilogue(b2_d.data_ptr(), 0.0, 0, 0, x_d.data_ptr() if N2 == K else 0, K * 4))
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d2)), "fc2 planes")
    ctx.sync()
    got_y, got_h = y.cpu().numpy(), h.cpu().numpy()
    for c in range(T):
        assert np.array_equal(got_y[c].view(np.uint32), alone_y[c].view(np.uint32)), (t, K, T, c)
        if only:
            assert (got_h[c] == 7.0).all()              # the F32 intermediate was never stored
        else:
            assert np.array_equal(got_h[c].view(np.uint32), alone_h[c].view(np.uint32))


@pytest.mark.parametrize("t", ["q5_0", "q4_K"])
@pytest.mark.parametrize("T,n_kv,H", [(3, 300, 20), (5, 1536, 20), (8, 129, 8), (2, 77, 4)])
def test_attention_combine_through_planes_is_bit_identical_to_the_fused_projection(gpu, oracle, t, T, n_kv, H):
    ctx, ka, torch = gpu
    tid = QT[t]
    D, K, N = 64, H * 64, 640
    if tid == 12 and K % 256:
        pytest.skip("Q4_K needs K % 256 == 0")
    rng = np.random.default_rng(T * 11 + n_kv + tid)
    q = (rng.standard_normal((T, H, D)) * 0.6).astype(np.float32)
    k = (rng.standard_normal((n_kv, H, D)) * 0.6).astype(np.float16)
    v = rng.standard_normal((n_kv, H, D)).astype(np.float16)
    _, planar = quantize(oracle, ka, tid, (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    res = rng.standard_normal((T, N)).astype(np.float32)
    q_d, k_d, v_d, w_d, b_d, r_d = (dev(torch, a) for a in (q, k, v, planar, bias, res))
    tq = ka.tensor(q_d.data_ptr(), ka.F32, [D, T, H], [4, H * D * 4, D * 4, T * H * D * 4])
    tk = ka.tensor(k_d.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
    tv = ka.tensor(v_d.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
    parts = ka.AttnPartials()
    ctx.check(ka.lib().mi355x_flash_attn_partial(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), None, 0.125, C.byref(parts)), "fattn_partial")
    ya = torch.zeros((T, N), dtype=torch.float32, device="cuda:0")
    yb = torch.zeros((T, N), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ep = ka.Epilogue(b_d.data_ptr(), 0.0, 0, 0, r_d.data_ptr(), N * 4)
    d = ka.GemvDesc()
    d.K, d.T, d.nseg = K, T, 1
    d.attn_part_o, d.attn_part_ml, d.attn_nparts = parts.part_o, parts.part_ml, parts.nparts
    _seg(ka, d, 0, w_d, tid, N, ya, ka.F32, ep)
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "fused combine + projection")
    planes = _planes(ctx, ka)
    a = ka.ActDesc()
    a.K, a.T, a.wtype = K, T, tid
    a.attn_part_o, a.attn_part_ml, a.attn_nparts = parts.part_o, parts.part_ml, parts.nparts
    ctx.check(ka.lib().mi355x_act_prepare(ctx.h, C.byref(a), planes), "act_prepare(combine)")
    d2 = ka.GemvDesc()
    d2.K, d2.T, d2.nseg, d2.x_planes = K, T, 1, planes
    _seg(ka, d2, 0, w_d, tid, N, yb, ka.F32, ep)
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d2)), "projection over planes")
    ctx.sync()
    assert np.array_equal(ya.cpu().numpy().view(np.uint32), yb.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("T", [8, 16, 20, 29])
@pytest.mark.parametrize("t", ["q5_0", "q8_0", "q4_K"])          # (Q4_K: no k_vocab — T > 8 goes image by image through k_gemv8)
def test_vocabulary_projection_over_planes_with_per_state_destinations(gpu, oracle, t, T):
    """final LayerNorm + logits (N = 51866: k_vocab) for T columns that belong to T states (T > 8: images of 8 columns)"""
    ctx, ka, torch = gpu
    tid = QT[t]
    K, N = 1280, 51866
    rng = np.random.default_rng(5 + tid)
    x = (rng.standard_normal((T, K)) * 2).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(K)).astype(np.float32)
    _, planar = quantize(oracle, ka, tid, (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    x_d, lw_d, lb_d, w_d = dev(torch, x), dev(torch, lw), dev(torch, lb), dev(torch, planar)
    alone = []
    for c in (0, 3, 7, T // 2 + 1, T - 1):
        y = torch.zeros((1, N), dtype=torch.float32, device="cuda:0")
        d = ka.GemvDesc()
        d.x, d.x_nb1, d.K, d.T, d.nseg, d.has_norm, d.eps, d.ln_w, d.ln_b = x_d.data_ptr() + c * K * 4, K * 4, K, 1, 1, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
        _seg(ka, d, 0, w_d, tid, N, y)
        ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "logits fused")
        ctx.sync()
        alone.append((c, y.cpu().numpy()[0]))
    planes = _planes(ctx, ka)
    a = ka.ActDesc()
    a.x, a.x_nb1, a.K, a.T, a.wtype, a.has_norm, a.eps, a.ln_w, a.ln_b = x_d.data_ptr(), K * 4, K, T, tid, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
    ctx.check(ka.lib().mi355x_act_prepare(ctx.h, C.byref(a), planes), "act_prepare")
    outs = [torch.zeros(N, dtype=torch.float32, device="cuda:0") for _ in range(T)]
    cols = ka.GemvCols()
    for c in range(T):
        cols.dst[0][c] = outs[c].data_ptr()
    d = ka.GemvDesc()
    d.K, d.T, d.nseg, d.x_planes, d.cols = K, T, 1, planes, C.addressof(cols)
    _seg(ka, d, 0, w_d, tid, N, None)
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "logits over planes")
    ctx.sync()
    for c, ref in alone:
        assert np.array_equal(outs[c].cpu().numpy().view(np.uint32), ref.view(np.uint32)), c


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("t,K,T", [("q5_0", 1280, 5), ("q5_0", 512, 3), ("q8_0", 1280, 5), ("q4_0", 768, 7), ("q5_0", 1280, 8), ("q5_0", 512, 2), ("q5_0", 1280, 1), ("q4_K", 1280, 5), ("q4_K", 1280, 20), ("q4_K", 512, 11), ("q5_0", 1280, 20)])
def test_vocabulary_projection_mirror_rows_equal_the_destination_rows(gpu, oracle, t, K, T, pinned):
    """k_vocab with a second (mirror) destination per column — the host-visible copy whisper's logits read-back is served from — for every
    column count incl. those below the kernel's built-in maximum (T = 3, 5, 7), planes form and LayerNorm form (T <= 2)"""
    ctx, ka, torch = gpu
    tid = QT[t]
    N = 51864 if K == 512 else 51866
    rng = np.random.default_rng(K + T + tid)
    x = (rng.standard_normal((T, K)) * 2).astype(np.float32)
    lw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(K)).astype(np.float32)
    _, planar = quantize(oracle, ka, tid, (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    x_d, lw_d, lb_d, w_d = dev(torch, x), dev(torch, lw), dev(torch, lb), dev(torch, planar)
    y = torch.zeros((T, N), dtype=torch.float32, device="cuda:0")
    # pinned: the mirror lives in page-locked HOST memory (what the backend uses), written by the kernel across PCIe
    mir = torch.full((T, N), -7.0, dtype=torch.float32).pin_memory() if pinned else torch.full((T, N), -7.0, dtype=torch.float32, device="cuda:0")
    cols = ka.GemvCols()
    for c in range(T):
        cols.dst[0][c] = y.data_ptr() + c * N * 4
        cols.mirror[c] = mir.data_ptr() + c * N * 4
    d = ka.GemvDesc()
    d.K, d.T, d.nseg, d.cols = K, T, 1, C.addressof(cols)
    _seg(ka, d, 0, w_d, tid, N, y)
    if T <= 2:
        d.x, d.x_nb1, d.has_norm, d.eps, d.ln_w, d.ln_b = x_d.data_ptr(), K * 4, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
    else:
        planes = _planes(ctx, ka)
        a = ka.ActDesc()
        a.x, a.x_nb1, a.K, a.T, a.wtype, a.has_norm, a.eps, a.ln_w, a.ln_b = x_d.data_ptr(), K * 4, K, T, tid, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
        ctx.check(ka.lib().mi355x_act_prepare(ctx.h, C.byref(a), planes), "act_prepare")
        d.x_planes = planes
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d)), "logits with mirror")
    assert ka.lib().mi355x_last_launch_mirrored(ctx.h) == 1
    ctx.sync()
    got, m = y.cpu().numpy(), mir.cpu().numpy().copy()
    assert np.isfinite(got).all() and np.abs(got).max() > 0.1
    for c in range(T):
        assert np.array_equal(got[c].view(np.uint32), m[c].view(np.uint32)), (t, K, T, c, int((got[c] != m[c]).sum()))


@pytest.mark.parametrize("S,H,kvs,masked", [(8, 20, (1, 17, 128, 129, 200, 255, 256, 77), True), (3, 8, (1536, 1536, 1536), False), (2, 6, (300, 5), True),
                                            (16, 20, tuple(range(3, 448, 28)), True), (19, 8, (1500,) * 19, False)])
def test_multi_state_attention_is_bit_identical_to_one_launch_per_state(gpu, S, H, kvs, masked):
    ctx, ka, torch = gpu
    D = 64
    rng = np.random.default_rng(S * 13 + H)
    n_ctx = max(kvs) + 3
    st = (ka.AttnState * S)()
    keep, singles = [], []
    for s in range(S):
        q = (rng.standard_normal((1, H, D)) * 0.6).astype(np.float32)
        k = (rng.standard_normal((n_ctx, H, D)) * 0.6).astype(np.float16)
        v = rng.standard_normal((n_ctx, H, D)).astype(np.float16)
        m = np.where(rng.random(n_ctx) < 0.2, -np.inf, 0.0).astype(np.float16)
        m[0] = 0
        q_d, k_d, v_d, m_d = (dev(torch, a) for a in (q, k, v, m))
        keep += [q_d, k_d, v_d, m_d]
        st[s].q, st[s].k, st[s].v, st[s].mask, st[s].n_kv = q_d.data_ptr(), k_d.data_ptr(), v_d.data_ptr(), (m_d.data_ptr() if masked else 0), kvs[s]
        tq = ka.tensor(q_d.data_ptr(), ka.F32, [D, 1, H], [4, H * D * 4, D * 4, H * D * 4])
        tk = ka.tensor(k_d.data_ptr(), ka.F16, [D, kvs[s], H], [2, H * D * 2, D * 2, n_ctx * H * D * 2])
        tv = ka.tensor(v_d.data_ptr(), ka.F16, [D, kvs[s], H], [2, H * D * 2, D * 2, n_ctx * H * D * 2])
        tm = ka.tensor(m_d.data_ptr(), ka.F16, [kvs[s], 1], [2, n_ctx * 2, n_ctx * 2, n_ctx * 2])
        parts = ka.AttnPartials()
        ctx.check(ka.lib().mi355x_flash_attn_partial(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), C.byref(tm) if masked else None, 0.125, C.byref(parts)), "single")
        o = torch.zeros((1, H, D), dtype=torch.float32, device="cuda:0")
        to = ka.tensor(o.data_ptr(), ka.F32, [D, H, 1])
        ctx.check(ka.lib().mi355x_flash_attn_combine(ctx.h, C.byref(parts), C.byref(to)), "combine")
        ctx.sync()
        singles.append(o.cpu().numpy()[0])
        if s == 0:
            shapes = (tq, tk, tv)
    parts = ka.AttnPartials()
    ctx.check(ka.lib().mi355x_flash_attn_partial_multi(ctx.h, S, st, C.byref(shapes[0]), C.byref(shapes[1]), C.byref(shapes[2]), 0.125, C.byref(parts)), "multi")
    assert parts.T == S and parts.H == H and parts.nparts == (max(kvs) + 127) // 128
    o = torch.zeros((S, H, D), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    to = ka.tensor(o.data_ptr(), ka.F32, [D, H, S])
    ctx.check(ka.lib().mi355x_flash_attn_combine(ctx.h, C.byref(parts), C.byref(to)), "combine")
    ctx.sync()
    got = o.cpu().numpy()
    for s in range(S):
        assert np.array_equal(got[s].view(np.uint32), singles[s].view(np.uint32)), s


@pytest.mark.parametrize("S,H,kvs,masked", [(8, 20, (1, 17, 128, 129, 200, 255, 256, 448), True), (5, 8, (33, 512, 64, 300, 2), True), (3, 6, (100, 100, 100), False),
                                            (8, 20, (1500,) * 8, False), (4, 8, (1500, 513, 1024, 1025), True), (3, 6, (1536, 7, 900), False),
                                            (16, 20, tuple(range(5, 448, 28)), True), (11, 8, (40,) * 11, True), (32, 6, tuple(range(1, 449, 14)), True)])
def test_self_attention_straight_to_planes_is_bit_identical_to_partials_combine_quantize(gpu, S, H, kvs, masked):
    """mi355x_flash_attn_planes (one launch; one, two or three rounds of 512 keys: self-attention and the 1500 keys of cross-attention) writes
    the same Q8_0 plane bytes as multi-state attention partials -> mi355x_act_prepare"""
    ctx, ka, torch = gpu
    D, K = 64, H * 64
    rng = np.random.default_rng(S * 17 + H)
    n_ctx = max(kvs) + 5
    st = (ka.AttnState * S)()
    keep = []
    for s in range(S):
        q = (rng.standard_normal((1, H, D)) * 0.6).astype(np.float32)
        k = (rng.standard_normal((n_ctx, H, D)) * 0.6).astype(np.float16)
        v = rng.standard_normal((n_ctx, H, D)).astype(np.float16)
        m = np.where(rng.random(n_ctx) < 0.2, -np.inf, 0.0).astype(np.float16)
        m[0] = 0
        q_d, k_d, v_d, m_d = (dev(torch, a) for a in (q, k, v, m))
        keep += [q_d, k_d, v_d, m_d]
        st[s].q, st[s].k, st[s].v, st[s].mask, st[s].n_kv = q_d.data_ptr(), k_d.data_ptr(), v_d.data_ptr(), (m_d.data_ptr() if masked else 0), kvs[s]
        if s == 0:
            tq = ka.tensor(q_d.data_ptr(), ka.F32, [D, 1, H], [4, H * D * 4, D * 4, H * D * 4])
            tk = ka.tensor(k_d.data_ptr(), ka.F16, [D, kvs[s], H], [2, H * D * 2, D * 2, n_ctx * H * D * 2])
            tv = ka.tensor(v_d.data_ptr(), ka.F16, [D, kvs[s], H], [2, H * D * 2, D * 2, n_ctx * H * D * 2])
    nbytes = ka.lib().mi355x_act_planes_bytes(ka.Q5_0, K, S)
    pa = torch.zeros(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    pb = torch.zeros(nbytes + 64, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    parts = ka.AttnPartials()
    ctx.check(ka.lib().mi355x_flash_attn_partial_multi(ctx.h, S, st, C.byref(tq), C.byref(tk), C.byref(tv), 0.125, C.byref(parts)), "partial_multi")
    a = ka.ActDesc()
    a.K, a.T, a.wtype = K, S, ka.Q5_0
    a.attn_part_o, a.attn_part_ml, a.attn_nparts = parts.part_o, parts.part_ml, parts.nparts
    ctx.check(ka.lib().mi355x_act_prepare(ctx.h, C.byref(a), pa.data_ptr()), "act_prepare(combine)")
    ctx.check(ka.lib().mi355x_flash_attn_planes(ctx.h, S, st, C.byref(tq), C.byref(tk), C.byref(tv), 0.125, pb.data_ptr()), "flash_attn_planes")
    ctx.sync()
    A, B = pa.cpu().numpy()[:nbytes], pb.cpu().numpy()[:nbytes]
    assert A.any() and np.array_equal(A, B), int((A != B).sum())


@pytest.mark.parametrize("t", ["q5_0", "q4_K", "f16"])
def test_multi_state_step_head_matches_get_rows_and_cast(gpu, oracle, t):
    ctx, ka, torch = gpu
    tid = ka.F16 if t == "f16" else QT[t]
    K, V, P, S = 1280, 999, 448, 5
    rng = np.random.default_rng(77)
    wf = (rng.standard_normal((V, K)) * 0.05).astype(np.float32)
    te = wf.astype(np.float16) if t == "f16" else quantize(oracle, ka, tid, wf)[1]
    pe = (rng.standard_normal((P, K)) * 0.02).astype(np.float32)
    te_d, pe_d = dev(torch, te), dev(torch, pe)
    tte = ka.tensor(te_d.data_ptr(), tid, [K, V])
    tpe = ka.tensor(pe_d.data_ptr(), ka.F32, [K, P])
    st = (ka.HeadState * S)()
    keep, want = [], []
    for s in range(S):
        tok, pos, n = int(rng.integers(0, V)), int(rng.integers(0, P)), int(rng.integers(1, 300))
        mask = np.where(rng.random(n) < 0.3, -np.inf, 0.0).astype(np.float32)
        tok_d, pos_d, m_d = dev(torch, np.array([tok], np.int32)), dev(torch, np.array([pos], np.int32)), dev(torch, mask)
        dst = torch.zeros(K, dtype=torch.float32, device="cuda:0")
        m16 = torch.zeros(n, dtype=torch.float16, device="cuda:0")
        ref = torch.zeros((1, K), dtype=torch.float32, device="cuda:0")
        ti = ka.tensor(tok_d.data_ptr(), ka.I32, [1]); tp = ka.tensor(pos_d.data_ptr(), ka.I32, [1]); tr = ka.tensor(ref.data_ptr(), ka.F32, [K, 1])
        ctx.check(ka.lib().mi355x_get_rows_add(ctx.h, C.byref(tte), C.byref(ti), C.byref(tpe), C.byref(tp), C.byref(tr)), "get_rows_add")
        keep += [tok_d, pos_d, m_d, dst, m16, ref]
        st[s].tok, st[s].pos, st[s].dst, st[s].mask_f32, st[s].mask_f16, st[s].n_mask = tok_d.data_ptr(), pos_d.data_ptr(), dst.data_ptr(), m_d.data_ptr(), m16.data_ptr(), n
        want.append((ref, dst, mask.astype(np.float16), m16))
    ctx.check(ka.lib().mi355x_decode_head_multi(ctx.h, S, st, C.byref(tte), C.byref(tpe)), "decode_head_multi")
    ctx.sync()
    for ref, dst, m_ref, m16 in want:
        assert np.array_equal(ref.cpu().numpy()[0].view(np.uint32), dst.cpu().numpy().view(np.uint32))
        assert np.array_equal(m_ref.view(np.uint16), m16.cpu().numpy().view(np.uint16))
    # a token id outside the table (the reference's get_rows asserts): that state's embedding is NaN, visibly wrong, never the stale vector
    # of the previous step; the other states of the launch are untouched
    bad = dev(torch, np.array([V + 7], np.int32))
    st[1].tok = bad.data_ptr()
    ctx.check(ka.lib().mi355x_decode_head_multi(ctx.h, S, st, C.byref(tte), C.byref(tpe)), "decode_head_multi")
    ctx.sync()
    assert np.isnan(want[1][1].cpu().numpy()).all()
    assert np.array_equal(want[0][0].cpu().numpy()[0].view(np.uint32), want[0][1].cpu().numpy().view(np.uint32))


# ---------------------------------------------------------------------------------------------------------------
# the whole thing through the unmodified reference host: S whisper_states on one device, one C++ thread each (native harness)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arch,qtype,streams", [("base.en", "q5_0", 4), ("base.en", "q4_k", 8), ("large-v3-2l", "q8_0", 8), ("base.en", "q5_0", 16), ("large-v3-2l", "q8_0", 11), ("large-v3-2l", "q4_k", 12), ("large-v3-2l", "q4_k", 30)])
def test_cross_state_batching_is_bit_identical_to_one_launch_chain_per_state(arch, qtype, streams):
    """every stream's final logits with the plugin's cross-state batching (the states' decode steps as the columns of one launch chain)
    equal, bit for bit, the same streams run with one launch chain per state — and merged chains did carry several columns"""
    from synth_model import make_model
    from whisper_cpp_amd import host_api as h
    m = make_model(arch, qtype)
    rows = {}
    for batching in (2, 0):                     # 2: merged chains from two decoding states on (1 would wait for five)
        r = h.run(m, use_gpu=True, n_devices=1, streams=streams, n_decode=24, steps=2, warmup=1, batching=batching)
        assert r["rc"] == 0 and r["error"] == "", r
        n = h.lib().mi355x_host_last_logits(None, 0)
        out = np.zeros(n, dtype=np.float32)
        assert h.lib().mi355x_host_last_logits(out.ctypes.data, out.size) == n and n % streams == 0
        rows[batching] = out.reshape(streams, -1).copy()
        if batching:
            st = r["batch_stats"]
            assert st["chains"] > 0 and st["columns"] > 1.5 * st["chains"] and st["fallbacks"] == 0, st
            if streams > 8:
                assert st["columns"] > 8 * st["chains"], st                 # chains of more than one image of 8 columns did run
        else:
            assert r["batch_stats"]["chains"] == 0, r["batch_stats"]
    h.run(m, use_gpu=True, n_devices=1, streams=1, n_decode=1, steps=1, warmup=0, batching=0)      # leave the switch off for whoever runs next
    assert np.isfinite(rows[2]).all()
    assert not np.array_equal(rows[2][0], rows[2][1])              # different audio per stream
    for s in range(streams):
        assert np.array_equal(rows[2][s].view(np.uint32), rows[0][s].view(np.uint32)), s


def test_argmax_top2_first_maximum_and_runner_up(gpu):
    ctx, ka, torch = gpu
    rng = np.random.default_rng(3)
    for n, plant in ((51866, None), (51864, (777, 40000)), (1000, (0, 999)), (70, None)):
        x = rng.standard_normal(n).astype(np.float32)
        if plant:                                   # two equal maxima: the first index must win, the margin is 0
            x[list(plant)] = 9.5
        x_d = dev(torch, x)
        out = torch.zeros(4, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        ctx.check(ka.lib().mi355x_argmax_top2(ctx.h, x_d.data_ptr(), n, out.data_ptr()), "argmax_top2")
        ctx.sync()
        o = out.cpu().numpy()
        idx, top1, top2 = int(o[0]), o[1:2].view(np.float32)[0], o[2:3].view(np.float32)[0]
        srt = np.sort(x)[::-1]
        assert idx == int(np.argmax(x)) and top1 == srt[0] and top2 == srt[1] and int(o[3]) == n, (n, idx, top1, top2)


@pytest.mark.parametrize("batching", [0, 2])
def test_device_side_greedy_sampling_equals_the_host_scan(batching):
    """free-running decode in the native harness: the token fed to the next step is the arg-max taken on the device
    (ggml_backend_mi355x_argmax_last); every one of them equals the host scan of the logits row whisper_decode returned"""
    from synth_model import make_model
    from whisper_cpp_amd import host_api as h
    m = make_model("base.en", "q5_0")
    r = h.run(m, use_gpu=True, n_devices=1, streams=3, n_decode=40, steps=1, warmup=1, batching=batching, device_greedy=True)
    assert r["rc"] == 0 and r["error"] == "", r
    assert r["greedy_checked"] == 3 * 40 * 2 and r["greedy_mismatches"] == 0, r
    h.run(m, use_gpu=True, n_devices=1, streams=1, n_decode=1, steps=1, warmup=0, batching=0)
