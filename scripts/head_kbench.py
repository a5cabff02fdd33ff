#!/usr/bin/env python3
"""The self-attention block of a single-token step, large-v3 shape (K = 1280, 20 heads, n_kv = 256): two launches (LN + Q/K/V, attention
partials) against the one-launch form (decode_head.hip), hipEvent-bracketed by the library's profiler; weights rotate through 32 layers."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from whisper_cpp_amd import kernels_api as ka  # noqa: E402


def main():
    import torch
    qt = sys.argv[1] if len(sys.argv) > 1 else "q5_0"
    tid = ka.TYPE_NAMES[qt]
    ctx = ka.Ctx(0)
    L = ka.lib()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    K, H, n_kv, n_ctx, NL = 1280, 20, 256, 512, 32
    N = H * 64

    def wq():
        nbytes = N * ka.row_bytes(tid, K)
        w = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda:0", generator=g)
        nblk = N * K // 32
        w[nbytes - nblk * 2:] = (torch.rand(nblk, device="cuda:0", generator=g) * 0.01).half().view(torch.uint8)
        return w
    ws = [[wq() for _ in range(3)] for _ in range(NL)]
    kcs = [torch.randn((n_ctx, N), device="cuda:0", generator=g).half() for _ in range(NL)]
    vcs = [torch.randn((n_ctx, N), device="cuda:0", generator=g).half() for _ in range(NL)]
    x = torch.randn(K, device="cuda:0", generator=g)
    lw, lb = torch.ones(K, device="cuda:0"), torch.zeros(K, device="cuda:0")
    bias = torch.zeros(N, device="cuda:0")
    q = torch.zeros(N, device="cuda:0")
    m = torch.zeros(n_kv, dtype=torch.float16, device="cuda:0")
    m[101:] = float("-inf")
    new_key = 100

    def desc(l):
        d = ka.GemvDesc()
        d.x, d.x_nb1, d.K, d.T, d.nseg, d.has_norm, d.eps, d.ln_w, d.ln_b = x.data_ptr(), K * 4, K, 1, 3, 1, 1e-5, lw.data_ptr(), lb.data_ptr()
        for s in range(3):
            d.seg[s].w, d.seg[s].wtype, d.seg[s].N = ws[l][s].data_ptr(), tid, N
            d.seg[s].ep.bias = bias.data_ptr() if s != 1 else None
            if s < 2:
                d.seg[s].ep.scale, d.seg[s].ep.has_scale = 0.35, 1
        d.seg[0].dst, d.seg[0].dst_type, d.seg[0].dst_nb1 = q.data_ptr(), ka.F32, N * 4
        d.seg[1].dst, d.seg[1].dst_type, d.seg[1].dst_nb1 = kcs[l].data_ptr() + new_key * N * 2, ka.F16, N * 2
        d.seg[2].dst, d.seg[2].dst_type, d.seg[2].dst_nb1 = vcs[l].data_ptr() + new_key * N * 2, ka.F16, N * 2
        return d
    descs = [desc(l) for l in range(NL)]
    tq = ka.tensor(q.data_ptr(), ka.F32, [64, 1, H], [4, N * 4, 256, N * 4])
    tks = [ka.tensor(kcs[l].data_ptr(), ka.F16, [64, n_kv, H], [2, N * 2, 128, n_ctx * N * 2]) for l in range(NL)]
    tvs = [ka.tensor(vcs[l].data_ptr(), ka.F16, [64, n_kv, H], [2, N * 2, 128, n_ctx * N * 2]) for l in range(NL)]
    tm = ka.tensor(m.data_ptr(), ka.F16, [n_kv, 1], [2, n_kv * 2, n_kv * 2, n_kv * 2])
    parts = ka.AttnPartials()

    def two(l):
        rc = L.mi355x_gemv_fused(ctx.h, C.byref(descs[l]))
        return rc or L.mi355x_flash_attn_partial(ctx.h, C.byref(tq), C.byref(tks[l]), C.byref(tvs[l]), C.byref(tm), 1.0, C.byref(parts))

    def one(l):
        return L.mi355x_self_attn_head(ctx.h, C.byref(descs[l]), 0, 1, 2, C.byref(tks[l]), C.byref(tvs[l]), C.byref(tm), 1.0, new_key, C.byref(parts))
    torch.cuda.synchronize()
    for name, fn in (("two launches", two), ("one launch", one), ("two launches", two), ("one launch", one)):
        for l in range(NL):
            rc = fn(l)
            assert rc == 0, (name, rc)
        ctx.sync()
        ctx.prof(True)
        ctx.prof_reset()
        for r in range(4):
            for l in range(NL):
                fn(l)
        ctx.sync()
        rows = ctx.prof_report()
        ctx.prof(False)
        tot = sum(r["total_ms"] for r in rows) * 1e3 / (4 * NL)
        print(f"{qt} {name:14s}: {tot:6.2f} us per block   " + "  ".join(f"{r['name'].split('(')[0][-40:]} {r['total_ms'] * 1e3 / max(r['calls'], 1):.2f}" for r in rows), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
