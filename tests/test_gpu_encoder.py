"""GPU tests of the round-3 encoder kernels (run with `-m gpu` on an MI355X), through the C ABI of include/mi355x_kernels.h:

  * k_fattn_mfma<NG, MASK>: NG key groups per workgroup (NG waves per SIMD, merged through LDS) and, without a mask, scale and
    log2(e) folded into one fma per score — every NG against an exact f64 attention on the same f16-rounded q (the bar of
    tests/test_gpu.py::test_flash_attn_vs_oracle: NMSE < 1e-6 against the truth) and against the NG = 1 form;
    (k_gemm_f16_ring's 256-row tiles and the grouped form's other tile shapes — measured slower in rounds 2 / 3 — were removed in round 5
    together with their switches and their bit-identity tests.)

The key-group count is forced through the kernel library's test hook (mi355x_test_option, include/mi355x_kernels.h), so one process can
compare the forms."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import nmse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from whisper_cpp_amd import kernels_api as ka
    ctx = ka.Ctx(0)
    yield ctx, ka, torch
    ctx.close()


class opt:
    """kernel library test option for the duration of a with-block (mi355x_test_option: unset again on exit)"""
    FATTN_NG = 3

    def __init__(self, ka, which, value):
        self.ka, self.which, self.value = ka, which, value

    def __enter__(self):
        self.ka.lib().mi355x_test_option(self.which, int(self.value), 1)

    def __exit__(self, *a):
        self.ka.lib().mi355x_test_option(self.which, 0, 0)


def dev(torch, a):
    t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("T,n_kv,H,mask", [(1500, 1500, 20, False), (1500, 1536, 4, False), (200, 320, 4, False), (130, 64, 3, False), (77, 1000, 2, False),
                                            (300, 300, 6, True), (256, 256, 8, True), (129, 705, 2, True)])
def test_attention_key_groups_agree_with_exact_attention_and_with_each_other(gpu, T, n_kv, H, mask):
    ctx, ka, torch = gpu
    D = 64
    rng = np.random.default_rng(T * 7 + n_kv + H)
    q = (rng.standard_normal((T, H, D)) * 0.6).astype(np.float32)
    q[:, 0, :] *= 6.0                                      # one head with scores of +-10: the running-maximum rescale matters
    k = (rng.standard_normal((n_kv, H, D)) * 0.6).astype(np.float16)
    v = rng.standard_normal((n_kv, H, D)).astype(np.float16)
    mf = None
    if mask:
        mf = np.zeros((T, n_kv), dtype=np.float32)
        for t in range(T):
            mf[t, max(1, n_kv - T + t + 1):] = -np.inf
    q_d, k_d, v_d = dev(torch, q), dev(torch, k), dev(torch, v)
    tq = ka.tensor(q_d.data_ptr(), ka.F32, [D, T, H], [4, H * D * 4, D * 4, T * H * D * 4])
    tk = ka.tensor(k_d.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
    tv = ka.tensor(v_d.data_ptr(), ka.F16, [D, n_kv, H], [2, H * D * 2, D * 2, n_kv * H * D * 2])
    tm = None
    if mask:
        m_d = dev(torch, mf.astype(np.float16))
        tm = C.byref(ka.tensor(m_d.data_ptr(), ka.F16, [n_kv, T]))
    qh = q.astype(np.float16).astype(np.float64)
    sc = np.einsum("thd,khd->htk", qh, k.astype(np.float64)) * 0.125
    if mask:
        sc = sc + mf.astype(np.float64)[None, :, :]
    sc = sc - sc.max(axis=-1, keepdims=True)
    pr = np.exp(sc)
    pr /= pr.sum(axis=-1, keepdims=True)
    exact = np.einsum("htk,khd->thd", pr, v.astype(np.float64))
    got = {}
    for ng in (1, 2, 3, 4):
        o_d = torch.full((T, H, D), 7.0, dtype=torch.float32, device="cuda:0")
        p_d = torch.zeros((T, H * D), dtype=torch.float16, device="cuda:0")
        p2_d = torch.zeros((T, H * D), dtype=torch.float16, device="cuda:0")
        torch.cuda.synchronize()
        with opt(ka, opt.FATTN_NG, ng):
            ctx.check(ka.lib().mi355x_flash_attn_ext_prep(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), tm, C.byref(ka.tensor(o_d.data_ptr(), ka.F32, [D, H, T])), 0.125, p_d.data_ptr()), "flash_attn_prep")
            ctx.sync()
        got[ng] = o_d.cpu().numpy()
        assert np.isfinite(got[ng]).all()
        e = nmse(exact, got[ng])
        assert e < 1e-6, (ng, e)
        # the activations written for the output projection are those of a separate pass over THIS result
        ctx.check(ka.lib().mi355x_prep_act(ctx.h, o_d.data_ptr(), H * D * 4, 0, p2_d.data_ptr(), H * D, T, 1), "prep_act")
        ctx.sync()
        assert np.array_equal(p_d.cpu().numpy().view(np.uint16), p2_d.cpu().numpy().view(np.uint16)), ng
    for ng in (2, 3, 4):
        # between the forms only the f16 rounding of P differs (it is taken against another running maximum): 2^-11 relative per weight
        assert nmse(got[1], got[ng]) < 1e-6, (ng, nmse(got[1], got[ng]))
        assert nmse(got[1][:, 0], got[ng][:, 0]) < 1e-6, ng   # per head too: the large-score head must not hide behind the others


@pytest.mark.parametrize("n0,n1,ld", [(1280, 1500, 1500), (384, 1500, 1500), (100, 77, 80), (64, 64, 64), (1280, 1500, 1504)])
def test_transposing_copy_is_exact(gpu, n0, n1, ld):
    """ggml_cont(ggml_transpose(x)) (src/whisper.cpp:2069): dst[i1][i0] = src[i0][i1], through the LDS-tile kernel"""
    ctx, ka, torch = gpu
    rng = np.random.default_rng(n0 + n1)
    a = rng.standard_normal((n0, ld)).astype(np.float32)
    a_d = dev(torch, a)
    d = torch.zeros((n1, n0), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ts = ka.tensor(a_d.data_ptr(), ka.F32, [n0, n1], [ld * 4, 4, n0 * ld * 4, n0 * ld * 4])       # the transposed view
    td = ka.tensor(d.data_ptr(), ka.F32, [n0, n1])
    n_before = ka.lib().mi355x_eager_count(ctx.h)
    ctx.check(ka.lib().mi355x_cpy(ctx.h, C.byref(ts), C.byref(td)), "cpy")
    ctx.sync()
    assert ka.lib().mi355x_eager_count(ctx.h) - n_before == 1
    assert np.array_equal(d.cpu().numpy(), a[:, :n1].T)


@pytest.mark.parametrize("M,K,T", [(3000, 384, 1280), (1500, 3840, 1280), (300, 192, 40)])
def test_gemm_epilogue_with_a_bias_per_column_and_gelu_equals_the_separate_ops(gpu, M, K, T):
    """the conv front end (src/whisper.cpp:2013-2020): mul_mat(im2col rows, kernel) + [1, OC] bias + GELU in the product's epilogue —
    word for word the f16-table GELU of (product + bias[column]) computed from the plain product"""
    ctx, ka, torch = gpu
    rng = np.random.default_rng(M + T)
    w = dev(torch, (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float16))
    act = dev(torch, rng.standard_normal((T, K)).astype(np.float16))
    bias = rng.standard_normal(T).astype(np.float32)
    bias_d = dev(torch, bias)
    tw = ka.tensor(w.data_ptr(), ka.F16, [K, M])
    y0 = torch.zeros((T, M), dtype=torch.float32, device="cuda:0")
    y1 = torch.zeros((T, M), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), K, T, y0.data_ptr(), M * 4, ka.F32, None), "gemm")
    ep = ka.Epilogue()
    ep.bias, ep.bias_per_col, ep.gelu = bias_d.data_ptr(), 1, 1
    ctx.check(ka.lib().mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), K, T, y1.data_ptr(), M * 4, ka.F32, C.byref(ep)), "gemm+ep")
    ctx.sync()
    tab = np.empty(65536, dtype=np.uint16)
    ka.lib().mi355x_gelu_table_host(tab.ctypes.data_as(C.c_void_p))
    x = y0.cpu().numpy() + bias[:, None]
    want = tab[x.astype(np.float16).view(np.uint16)].view(np.float16).astype(np.float32)
    want = np.where(x <= -10.0, 0.0, np.where(x >= 10.0, x, want)).astype(np.float32)
    got = y1.cpu().numpy()
    assert np.abs(got).max() > 0.1 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # a per-column bias on the mat-vec paths is refused, not mis-applied
    x8 = dev(torch, rng.standard_normal((4, K)).astype(np.float32))
    y8 = torch.zeros((4, M), dtype=torch.float32, device="cuda:0")
    rc = ka.lib().mi355x_mul_mat(ctx.h, C.byref(tw), C.byref(ka.tensor(x8.data_ptr(), ka.F32, [K, 4])), C.byref(ka.tensor(y8.data_ptr(), ka.F32, [M, 4])), C.byref(ep))
    assert rc in (-1, 0)
    if rc == 0:                     # served by the MFMA path (the mat-vec kernels refuse the flag): right values
        ctx.sync()
        acc = x8.cpu().numpy().astype(np.float16).astype(np.float64) @ w.cpu().numpy().astype(np.float64).T + bias[:4, None]
        ref = 0.5 * acc * (1.0 + np.tanh(0.7978845608028654 * acc * (1.0 + 0.044715 * acc * acc)))
        assert np.abs(y8.cpu().numpy() - ref).max() < 5e-3


@pytest.mark.parametrize("M,K,T", [(1280, 1280, 1500), (384, 384, 200), (516, 256, 77)])
def test_gemm_with_an_f16_destination_equals_the_rounded_f32_result(gpu, M, K, T):
    """the K / V projections write F16 (ggml_cpy folded into the product): four halves per store on the vector path, the same
    round-to-nearest-even of the same f32 values as the F32 product followed by a cast (M = 516: the scalar path, M % 4 == 0 but rows not 16-byte aligned)"""
    ctx, ka, torch = gpu
    rng = np.random.default_rng(M + T)
    w = dev(torch, (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float16))
    act = dev(torch, rng.standard_normal((T, K)).astype(np.float16))
    bias = dev(torch, rng.standard_normal(M).astype(np.float32))
    ep = ka.Epilogue()
    ep.bias, ep.scale, ep.has_scale = bias.data_ptr(), 0.35, 1
    tw = ka.tensor(w.data_ptr(), ka.F16, [K, M])
    y32 = torch.zeros((T, M), dtype=torch.float32, device="cuda:0")
    y16 = torch.zeros((T, M), dtype=torch.float16, device="cuda:0")
    torch.cuda.synchronize()
    ctx.check(ka.lib().mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), K, T, y32.data_ptr(), M * 4, ka.F32, C.byref(ep)), "gemm f32")
    ctx.sync()
    ctx.check(ka.lib().mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), K, T, y16.data_ptr(), M * 2, ka.F16, C.byref(ep)), "gemm f16")
    ctx.sync()
    want = y32.cpu().numpy().astype(np.float16)
    got = y16.cpu().numpy()
    assert np.abs(want.astype(np.float32)).max() > 0.1 and np.array_equal(got.view(np.uint16), want.view(np.uint16))


# ---------------------------------------------------------------------------------------------------------------
# k_gemm_dq (round 6, GGML_MI355X_MMQ=3): quantized weight planes unpacked per workgroup into LDS + f16 activations by LDS-DMA, f16 MFMA
# ---------------------------------------------------------------------------------------------------------------
DQ_GEMM, DQ_BN = 4, 5          # mi355x_kernels.h: mi355x_test_option values
QT = {"q4_0": 2, "q5_0": 6, "q8_0": 8, "q4_K": 12}


def _rand_planar(torch, ka, tid, N, K, seed):
    """random weight planes (any bit pattern with finite scales is a legal block) in the kernel library's planar layout"""
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    nbytes = N * ka.row_bytes(tid, K)
    w = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
    if tid == 12:
        nsb = N * K // 256
        w[nbytes - nsb * 4:] = (torch.rand(nsb * 2, device="cuda:0", generator=g) * 0.02 + 0.001).half().view(torch.uint8)
    else:
        nblk = N * K // 32
        w[nbytes - nblk * 2:] = ((torch.rand(nblk, device="cuda:0", generator=g) - 0.5) * 0.04).half().view(torch.uint8)
    return w


@pytest.mark.parametrize("t", list(QT))
@pytest.mark.parametrize("K,N,T", [(1280, 1280, 1500), (1280, 320, 333), (5120, 256, 130), (512, 384, 64)])
def test_dequant_to_lds_gemm_equals_the_register_staged_kernel_and_the_ring_on_the_f16_copy(gpu, t, K, N, T):
    """k_gemm_dq against (a) k_gemm_mfma, the register-staged kernel that dequantizes the same planes in its loop (MI355X_OPT_DQ_GEMM = 0) and
    (b) k_gemm_f16_ring on mi355x_dequant_f16's copy of the weight: the same f16 values, the same fragment layout, MFMA and K order — every
    output word equal, for both token-tile widths (128 / 256 columns: 4 / 8 waves), ragged last tiles in both dimensions (T = 1500, 333, 130;
    N = 320), bias + scale + residual epilogues and F16 destinations.  That the values ARE the reference's dequantized weights is
    tests/test_gpu.py::test_f16_weight_copy_is_bit_identical_to_fused_dequant; against the oracle's mul_mat: test_mul_mat_vs_oracle."""
    ctx, ka, torch = gpu
    tid = QT[t]
    if tid == 12 and K % 256:
        pytest.skip("Q4_K needs K % 256 == 0")
    L = ka.lib()
    w = _rand_planar(torch, ka, tid, N, K, 5 + tid + K)
    g = torch.Generator(device="cuda:0").manual_seed(99)
    act = torch.randn((T, K), device="cuda:0", generator=g).half()
    bias = torch.randn(N, device="cuda:0", generator=g)
    res = torch.randn((T, N), device="cuda:0", generator=g)
    sh = torch.zeros((N, K), dtype=torch.float16, device="cuda:0")
    tw = ka.tensor(w.data_ptr(), tid, [K, N])
    ctx.check(L.mi355x_dequant_f16(ctx.h, C.byref(tw), sh.data_ptr()), "dequant")
    ts = ka.tensor(sh.data_ptr(), ka.F16, [K, N])
    ep = ka.Epilogue(bias.data_ptr(), 0.5, 1, 0, res.data_ptr(), N * 4)

    def run(tens, f16_dst):
        y = torch.zeros((T, N), dtype=torch.float16 if f16_dst else torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        ctx.check(L.mi355x_gemm_f16act(ctx.h, C.byref(tens), act.data_ptr(), K, T, y.data_ptr(), N * (2 if f16_dst else 4), ka.F16 if f16_dst else ka.F32,
                                       None if f16_dst else C.byref(ep)), "gemm")
        ctx.sync()
        return y.cpu().numpy()

    for f16_dst in (False, True):
        ring = run(ts, f16_dst)
        staged = run(tw, f16_dst)                      # (MI355X_OPT_DQ_GEMM unset: the register-staged kernel)
        assert np.isfinite(ring.astype(np.float32)).all() and np.abs(ring.astype(np.float32)).max() > 1e-3
        view = np.uint16 if f16_dst else np.uint32
        assert np.array_equal(staged.view(view), ring.view(view))
        for bn in (128, 256):
            with opt(ka, DQ_GEMM, 1), opt(ka, DQ_BN, bn):
                got = run(tw, f16_dst)
            bad = np.argwhere(got.view(view) != ring.view(view))
            assert bad.size == 0, (bn, f16_dst, bad[:4].tolist(), len(bad))


def test_grouped_dequant_to_lds_gemm_and_its_activation_writing_epilogue(gpu):
    """the Q / K / V group of an encoder layer at full size (three members on the same activations, one launch, 8-wave tiles) equals the
    three single launches word for word; fc1 + bias + GELU whose epilogue leaves fc2's f16(d*q) activations equals the two-pass form
    (mi355x_gemm_f16act + mi355x_prep_act) — both through k_gemm_dq"""
    ctx, ka, torch = gpu
    L = ka.lib()
    tid, K, N, T = 6, 1280, 1280, 1500
    g = torch.Generator(device="cuda:0").manual_seed(3)
    ws = [_rand_planar(torch, ka, tid, N, K, 40 + i) for i in range(3)]
    act = torch.randn((T, K), device="cuda:0", generator=g).half()
    bias = torch.randn(N, device="cuda:0", generator=g)

    def launch(i, y):
        ep = ka.Epilogue(bias.data_ptr() if i != 1 else None, 0.25, 1 if i == 1 else 0)
        ctx.check(L.mi355x_gemm_f16act(ctx.h, C.byref(ka.tensor(ws[i].data_ptr(), tid, [K, N])), act.data_ptr(), K, T, y.data_ptr(), N * 4, ka.F32, C.byref(ep)), "gemm")

    single = [torch.zeros((T, N), device="cuda:0") for _ in range(3)]
    grouped = [torch.zeros((T, N), device="cuda:0") for _ in range(3)]
    torch.cuda.synchronize()
    L.mi355x_test_option(DQ_GEMM, 1, 1)
    for i in range(3):
        launch(i, single[i]); ctx.sync()
    n0 = L.mi355x_eager_count(ctx.h)
    for i in range(3):
        launch(i, grouped[i])
    ctx.sync()
    assert L.mi355x_eager_count(ctx.h) - n0 == 1
    for a, b in zip(single, grouped):
        assert torch.isfinite(a).all() and float(a.abs().max()) > 1e-3 and torch.equal(a.view(torch.int32), b.view(torch.int32))
    # fc1 -> fc2's activations
    M = 5120
    w1 = _rand_planar(torch, ka, tid, M, K, 77)
    b1 = torch.randn(M, device="cuda:0", generator=g) * 0.1
    ep = ka.Epilogue(b1.data_ptr(), 0.0, 0, 1)
    y = torch.zeros((T, M), device="cuda:0"); p2 = torch.zeros((T, M), dtype=torch.float16, device="cuda:0"); p1 = torch.zeros_like(p2)
    torch.cuda.synchronize()
    tw = ka.tensor(w1.data_ptr(), tid, [K, M])
    ctx.check(L.mi355x_gemm_f16act(ctx.h, C.byref(tw), act.data_ptr(), K, T, y.data_ptr(), M * 4, ka.F32, C.byref(ep)), "fc1")
    ctx.check(L.mi355x_prep_act(ctx.h, y.data_ptr(), M * 4, 0, p1.data_ptr(), M, T, 1), "prep")
    ctx.check(L.mi355x_gemm_f16act_prep(ctx.h, C.byref(tw), act.data_ptr(), K, T, None, M * 4, C.byref(ep), p2.data_ptr()), "fc1 + prep")
    ctx.sync()
    L.mi355x_test_option(DQ_GEMM, 0, 0)
    assert float(p1.float().abs().max()) > 1e-3 and torch.equal(p1.view(torch.int16), p2.view(torch.int16))
