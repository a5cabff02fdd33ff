"""GPU tests of the one-launch self-attention block of a single-token decoder step (decode_head.hip, mi355x_self_attn_head):
LayerNorm -> Q / K / V projections (+ bias / scale) -> KV-cache rows -> attention, one workgroup per head.  It must leave
exactly what the two launches it replaces leave — mi355x_gemv_fused (LayerNorm + three segments, K / V stored as F16 into the
caches) followed by mi355x_flash_attn_partial: the query vector, the cache rows and the combined attention output, bit for bit.
The attention result is additionally held against an f64 attention over the same cache contents."""
import ctypes as C

import numpy as np
import pytest

from conftest import nmse
from test_gpu import QT, dev, gpu, quantize  # noqa: F401  (gpu is a fixture)
from test_gpu_batch import _seg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("t", ["q4_0", "q5_0", "q8_0"])
@pytest.mark.parametrize("K,H,n_kv,new_key,masked", [(1280, 20, 256, 0, True), (1280, 20, 256, 131, True), (1280, 20, 256, 255, True), (1280, 20, 512, 300, True),
                                                      (384, 6, 256, 17, True), (512, 8, 64, 63, False), (1024, 16, 160, 5, True), (1280, 20, 32, 31, True)])
def test_one_launch_self_attention_equals_the_two_launches(gpu, oracle, t, K, H, n_kv, new_key, masked):
    ctx, ka, torch = gpu
    tid = QT[t]
    N = H * 64
    assert N == K
    rng = np.random.default_rng(K + n_kv + new_key + tid)
    x = (rng.standard_normal(K) * 2.0 + 0.3).astype(np.float32)
    lw, lb = rng.standard_normal(K).astype(np.float32), (rng.standard_normal(K) * 0.1).astype(np.float32)
    ws = []
    for _ in range(3):
        wf = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        _, planar = quantize(oracle, ka, tid, wf)
        ws.append(dev(torch, planar))
    bq, bv = dev(torch, rng.standard_normal(N).astype(np.float32)), dev(torch, rng.standard_normal(N).astype(np.float32))
    n_ctx = 512
    kc0 = (rng.standard_normal((n_ctx, N)) * 0.6).astype(np.float16)
    vc0 = rng.standard_normal((n_ctx, N)).astype(np.float16)
    m = np.zeros(n_kv, dtype=np.float16)
    m[new_key + 1:] = -np.inf                                            # cells past the step's own are not this sequence's
    if new_key > 4:
        m[2] = -np.inf                                                   # ... and one hole inside
    x_d, lw_d, lb_d, m_d = dev(torch, x), dev(torch, lw), dev(torch, lb), dev(torch, m)
    scale = float(64 ** -0.25)

    def desc(q_d, kc_d, vc_d):
        d = ka.GemvDesc()
        d.x, d.x_nb1, d.K, d.T, d.nseg, d.has_norm, d.eps, d.ln_w, d.ln_b = x_d.data_ptr(), K * 4, K, 1, 3, 1, 1e-5, lw_d.data_ptr(), lb_d.data_ptr()
        eq, ek, ev = ka.Epilogue(), ka.Epilogue(), ka.Epilogue()
        eq.bias, eq.scale, eq.has_scale = bq.data_ptr(), scale, 1          # src/whisper.cpp:2567-2584: Q = (Wq x + b) * s, K = (Wk x) * s, V = Wv x + b
        ek.scale, ek.has_scale = scale, 1
        ev.bias = bv.data_ptr()
        _seg(ka, d, 0, ws[0], tid, N, q_d, ep=eq)
        _seg(ka, d, 1, ws[1], tid, N, None, dst_type=ka.F16, ep=ek)
        _seg(ka, d, 2, ws[2], tid, N, None, dst_type=ka.F16, ep=ev)
        d.seg[1].dst = kc_d.data_ptr() + new_key * N * 2                  # this step's rows in the caches
        d.seg[2].dst = vc_d.data_ptr() + new_key * N * 2
        return d, (eq, ek, ev)

    def views(q_d, kc_d, vc_d):
        tq = ka.tensor(q_d.data_ptr(), ka.F32, [64, 1, H], [4, N * 4, 256, N * 4])
        tk = ka.tensor(kc_d.data_ptr(), ka.F16, [64, n_kv, H], [2, N * 2, 128, n_ctx * N * 2])
        tv = ka.tensor(vc_d.data_ptr(), ka.F16, [64, n_kv, H], [2, N * 2, 128, n_ctx * N * 2])
        tm = ka.tensor(m_d.data_ptr(), ka.F16, [n_kv, 1], [2, n_kv * 2, n_kv * 2, n_kv * 2])
        return tq, tk, tv, tm

    def combine(parts):
        o = torch.zeros((1, H, 64), dtype=torch.float32, device="cuda:0")
        ctx.check(ka.lib().mi355x_flash_attn_combine(ctx.h, C.byref(parts), C.byref(ka.tensor(o.data_ptr(), ka.F32, [64, H, 1]))), "combine")
        ctx.sync()
        return o.cpu().numpy()[0]

    # two launches
    q1, kc1, vc1 = torch.zeros(N, dtype=torch.float32, device="cuda:0"), dev(torch, kc0), dev(torch, vc0)
    d1, keep1 = desc(q1, kc1, vc1)
    ctx.check(ka.lib().mi355x_gemv_fused(ctx.h, C.byref(d1)), "gemv_fused")
    tq, tk, tv, tm = views(q1, kc1, vc1)
    p1 = ka.AttnPartials()
    ctx.check(ka.lib().mi355x_flash_attn_partial(ctx.h, C.byref(tq), C.byref(tk), C.byref(tv), C.byref(tm) if masked else None, 1.0, C.byref(p1)), "partial")
    o1 = combine(p1)
    # one launch (K / V / Q handed over in another segment order than q, k, v: the mapping is an argument)
    q2, kc2, vc2 = torch.zeros(N, dtype=torch.float32, device="cuda:0"), dev(torch, kc0), dev(torch, vc0)
    d2, keep2 = desc(q2, kc2, vc2)
    tq, tk, tv, tm = views(q2, kc2, vc2)
    p2 = ka.AttnPartials()
    rc = ka.lib().mi355x_self_attn_head(ctx.h, C.byref(d2), 0, 1, 2, C.byref(tk), C.byref(tv), C.byref(tm) if masked else None, 1.0, new_key, C.byref(p2))
    ctx.check(rc, "self_attn_head")
    assert p2.nparts == p1.nparts == (n_kv + 127) // 128 and p2.H == H and p2.T == 1
    o2 = combine(p2)
    assert np.isfinite(o1).all() and np.abs(o1).max() > 1e-3
    assert np.array_equal(q1.cpu().numpy().view(np.uint32), q2.cpu().numpy().view(np.uint32))
    k1, k2, v1, v2 = (a.cpu().numpy() for a in (kc1, kc2, vc1, vc2))
    assert np.array_equal(k1.view(np.uint16), k2.view(np.uint16)) and np.array_equal(v1.view(np.uint16), v2.view(np.uint16))
    assert not np.array_equal(k1[new_key].view(np.uint16), kc0[new_key].view(np.uint16))      # the step's row was written
    assert np.array_equal(np.delete(k1, new_key, 0).view(np.uint16), np.delete(kc0, new_key, 0).view(np.uint16))   # and nothing else
    assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32))
    # exact attention over the cache as it now is
    qh = q2.cpu().numpy().astype(np.float16).astype(np.float64).reshape(H, 64)
    kk = k2[:n_kv].astype(np.float64).reshape(n_kv, H, 64)
    vv = v2[:n_kv].astype(np.float64).reshape(n_kv, H, 64)
    sc = np.einsum("hd,khd->hk", qh, kk) + (m.astype(np.float64)[None, :] if masked else 0.0)
    sc = sc - sc.max(axis=-1, keepdims=True)
    pr = np.exp(sc)
    pr /= pr.sum(axis=-1, keepdims=True)
    exact = np.einsum("hk,khd->hd", pr, vv)
    assert nmse(exact, o2) < 1e-9
