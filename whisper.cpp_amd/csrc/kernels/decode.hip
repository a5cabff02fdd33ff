// Decoder-step kernels, second generation (T <= 8 columns): the launch/latency-bound half of the hot path.
//
// One whisper decode step is ~8 all-to-all dependent phases per layer (LN+QKV, self-attention, O-proj, LN+Q_cross,
// cross-attention, O-proj, LN+fc1+GELU, fc2), each only 1-8 MB of HBM traffic, i.e. each phase costs one memory
// round trip plus a kernel boundary rather than bytes/bandwidth.  These kernels are built around that:
//   k_gemv8      : quantized mat-vec.  8 lanes per weight row, 8 rows per wave pass => every lane busy for the
//                  row lengths Whisper uses (nb = K/32 = 16/24/32/40/160 blocks), 128-byte contiguous segments per
//                  row per load instruction, 3-step instead of 6-step lane reduction.  The first chunk of weight
//                  loads is issued BEFORE the activation prologue (LayerNorm / attention-combine / Q8_0
//                  quantization into LDS), so the HBM latency of the weights overlaps the prologue instead of
//                  following it.  Non-temporal loads: every weight byte is used exactly once.
//   k_fattn_dec  : decode attention, 128 keys per workgroup (4 waves x 32 keys), all K and V loads of a wave in
//                  flight at once (8 x 16 B per lane), in-wave softmax, one partial (m, l, o[64]) record per
//                  (head, query, 128-key chunk).
//   the combine of those partial records is folded into the prologue of the mat-vec that consumes the attention
//   output (the O-projection), so attention costs ONE kernel instead of two; k_fattn_combine2 is the stand-alone
//   fallback when no such consumer follows.
//
// Reference arithmetic reproduced: activations -> Q8_0 blocks (arch/x86/quants.c:302-398), integer block dot products,
// f32 accumulation of d_w*d_x*isum (ggml-cpu/quants.c:225-259, :365-406, :451-479); flash_attn_ext semantics of
// ggml-cpu/ops.cpp:8479-8715 (q rounded to f16, f32 scores, online softmax; V accumulated in f32 here, f16 there).
#include "decode_common.h"

template <int WT, int T, int LPR>
__global__ void __launch_bounds__(256) k_gemv8(const DGArgs a) {
    constexpr bool Q4K = WT == MI355X_TYPE_Q4_K;                       // lane-units: 64-element chunks instead of 32-element blocks
    constexpr int U = Q4K ? (LPR == 8 ? 3 : (LPR == 16 ? 2 : 1)) : DG_U(LPR), RPW = 64 / LPR;      // units per lane per chunk, rows per wave pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int wave = tid >> 6, lane = tid & 63;
    const int K = a.K, nb = Q4K ? K >> 6 : K >> 5, nsb = K >> 8;
    const int r8 = lane / LPR, j8 = lane % LPR;
    const int ntot = a.row_start[a.nseg];
    const int nchunks = (nb + LPR*U - 1) / (LPR*U);
    const int total = a.passes * nchunks;
    const int wg = blockIdx.x * nwaves + wave;

    // ---- per-lane row bookkeeping for pass p: returns weight base / block count of this lane's row ----
    auto row_of = [&](int pass, int & s, int & row) -> bool {
        const int grow = (wg * a.passes + pass) * RPW + r8;
        s = 0;
        if (a.nseg > 1 && grow >= a.row_start[1]) s = 1;
        if (a.nseg > 2 && grow >= a.row_start[2]) s = 2;
        row = grow - a.row_start[s];
        return grow < ntot;
    };
    auto load_chunk = [&](wblk<WT> * r, int it) {
        const int pass = it / nchunks, c = it - pass*nchunks;
        int s, row; const bool rok = row_of(pass, s, row);
        const char * base = (const char *) (s == 0 ? a.seg[0].w : (s == 1 ? a.seg[1].w : a.seg[2].w));
        const int64_t nbt = s == 0 ? a.seg[0].nbt : (s == 1 ? a.seg[1].nbt : a.seg[2].nbt);
        #pragma unroll
        for (int u = 0; u < U; u++) {
            const int g = j8 + LPR*(c*U + u);
            if constexpr (Q4K) wblk_load_q4k(r[u], base, nbt, rok ? row : 0, nsb, g < nb ? g : nb - 1);
            else               wblk_load<WT>(r[u], base, nbt, (int64_t) (rok ? row : 0) * nb + (g < nb ? g : nb - 1));      // clamped, never predicated
        }
    };

    // ---- prologue: activations -> [combine attention partials] -> [LayerNorm] -> Q8_0 planes in LDS ----
    float * red = (float *) smem;
    uint32_t * lo = (uint32_t *) (smem + 256);
    uint32_t * hi = lo + (size_t) T*nb*4;
    float * dx = Q4K ? (float *) (smem + 256 + (size_t) T*K) : (float *) (hi + (size_t) T*nb*4);
    int *   sx = Q4K ? (int *) (dx + T*nsb) : (int *) (dx + T*nb);
    float * stage = Q4K ? (float *) (smem + 256 + ((((size_t) T * ((size_t) K + nsb*4 + (K >> 5)*4)) + 15) & ~(size_t) 15)) : (float *) (sx + T*nb);
    const bool staged = a.x == nullptr || a.has_norm;
    const int K4 = K >> 2;
    // 4 consecutive values of column t starting at element e -> quantized activation planes (Q8_K: whole waves call this
    // together, one wave = one 256-element super-block: K % 256 == 0 and the loops below advance by whole waves)
    auto act_store = [&](const float v[4], int e, int t) {
        if constexpr (Q4K) dg_q8_K_store(v, e, t, K, T, lo, dx, sx);
        else               dg_q8_0_store(v, e, t, nb, lo, hi, dx, sx);
    };

    // Loads return in issue order (vmcnt), so the short-latency activation loads (L2 hits) go first and the first chunk
    // of weights (HBM misses) right behind them: the prologue then runs while the weights are still in flight.
    // xfirst: the whole activation fits DG_XR float4 registers per thread; otherwise the weights are requested first and
    // the activation streams through behind them.
    wblk<WT> cur[U], nxt[U];
    float4 xr[DG_XR];
    if (a.xfirst) {
        #pragma unroll
        for (int i = 0; i < DG_XR; i++) {
            const int idx = tid + i*nthreads;
            const int t = idx / K4, e4 = idx - t*K4;
            xr[i] = idx < T*K4 ? *(const float4 *) ((const char *) a.x + (int64_t) t*a.x_nb1 + (int64_t) e4*16) : make_float4(0, 0, 0, 0);
        }
    }
    load_chunk(cur, 0);
    if (total > 1) load_chunk(nxt, 1);

    if (a.xq) {
        // activations arrive quantized (decode_q.hip: k_act_prepare / a producer's epilogue): the image of the planes is copied as it is
        const int n16 = (int) ((dg_act_bytes(WT, K, T) + 15) >> 4);
        for (int idx = tid; idx < n16; idx += nthreads) ((uint4 *) (smem + 256))[idx] = ((const uint4 *) a.xq)[idx];
    } else if (a.xfirst) {
        if (a.has_norm) {
            #pragma unroll
            for (int i = 0; i < DG_XR; i++) {
                const int idx = tid + i*nthreads;
                if (idx < T*K4) *(float4 *) (stage + (size_t) idx*4) = xr[i];          // stage[t][K] is contiguous: idx*4 == t*K + e4*4
            }
            __syncthreads();
        } else {
            #pragma unroll
            for (int i = 0; i < DG_XR; i++) {
                const int idx = tid + i*nthreads;
                if (idx < T*K4) {
                    const int t = idx / K4, e4 = idx - t*K4;
                    const float v[4] = { xr[i].x, xr[i].y, xr[i].z, xr[i].w };
                    act_store(v, e4*4, t);
                }
            }
        }
    } else if (a.x == nullptr) {
        // x[t][h*64 + d] = sum_p w_p o_p[d] / sum_p w_p l_p,  w_p = exp(m_p - max_p m_p)   (k_fattn_dec records)
        for (int idx = tid; idx < T*K4; idx += nthreads) {
            const int t = idx / K4, e4 = idx - t*K4, h = e4 >> 4, d = (e4 & 15) << 2;
            const int64_t base = ((int64_t) h*T + t) * a.nparts;
            float M = -1e30f;
            for (int p = 0; p < a.nparts; p++) M = fmaxf(M, a.part_ml[(base + p)*2]);
            float L = 0.0f; float4 o = make_float4(0, 0, 0, 0);
            for (int p = 0; p < a.nparts; p++) {
                const float2 ml = *(const float2 *) (a.part_ml + (base + p)*2);
                const float w = __expf(ml.x - M);
                const float4 v = *(const float4 *) (a.part_o + (base + p)*64 + d);
                L = fmaf(w, ml.y, L);
                o.x = fmaf(w, v.x, o.x); o.y = fmaf(w, v.y, o.y); o.z = fmaf(w, v.z, o.z); o.w = fmaf(w, v.w, o.w);
            }
            const float inv = L == 0.0f ? 0.0f : 1.0f / L;
            *(float4 *) (stage + (size_t) t*K + e4*4) = make_float4(o.x*inv, o.y*inv, o.z*inv, o.w*inv);
        }
        __syncthreads();
    } else if (a.has_norm) {
        for (int idx = tid; idx < T*K4; idx += nthreads) {
            const int t = idx / K4, e4 = idx - t*K4;
            *(float4 *) (stage + (size_t) t*K + e4*4) = *(const float4 *) ((const char *) a.x + (int64_t) t*a.x_nb1 + (int64_t) e4*16);
        }
        __syncthreads();
    }
    if (a.xq) {
    } else if (a.has_norm) {
        // ggml_norm (ggml-cpu/ops.cpp:3698-3765) + affine: mean, then variance of (x - mean), y = (x-mean)*rsqrt(var+eps)*w + b
        float part[T];
        // K <= 2048: the reduction TREE of k_act_prepare MODE 1 (decode_q.hip) — element e4 belongs to "virtual wave" e4 / 64, whose 64
        // float4 are summed by one wave_sum, and the (up to 8) wave sums are added in wave order — so that a column normalised here
        // (own chain, T = 1) and in a cross-state batch (k_act_prepare -> planes) carries the same bits (round 4: the strided per-thread
        // sums used before differed from it in the last place for some columns: Q4_K vocabulary projection, K = 1280).  Larger K keeps
        // the strided form (no plane pipeline there).
        const bool vtree = K4 <= 512;
        const int nvw = (K4 + 63) >> 6;
        #pragma unroll
        for (int t = 0; t < T; t++) {
            if (vtree) {
                for (int vw = wave; vw < nvw; vw += nwaves) {
                    const int e4 = vw*64 + lane;
                    float p = 0.0f;
                    if (e4 < K4) { const float4 v = *(const float4 *) (stage + (size_t) t*K + e4*4); p += (v.x + v.y) + (v.z + v.w); }
                    p = wave_sum(p);
                    if (lane == 0) red[vw*8 + t] = p;
                }
            } else {
                float s = 0.0f;
                for (int e4 = tid; e4 < K4; e4 += nthreads) { const float4 v = *(const float4 *) (stage + (size_t) t*K + e4*4); s += (v.x + v.y) + (v.z + v.w); }
                s = wave_sum(s);
                if (lane == 0) red[wave*8 + t] = s;
            }
        }
        __syncthreads();
        float mean[T];
        #pragma unroll
        for (int t = 0; t < T; t++) {
            float s = 0.0f;
            if (vtree) { for (int w = 0; w < 8; w++) { const float pw = red[(w < nvw ? w : 0)*8 + t]; s += w < nvw ? pw : 0.0f; } }
            else for (int w = 0; w < nwaves; w++) s += red[w*8 + t];
            mean[t] = s / K;
        }
        __syncthreads();
        #pragma unroll
        for (int t = 0; t < T; t++) {
            if (vtree) {
                for (int vw = wave; vw < nvw; vw += nwaves) {
                    const int e4 = vw*64 + lane;
                    float p = 0.0f;
                    if (e4 < K4) {
                        const float4 v = *(const float4 *) (stage + (size_t) t*K + e4*4);
                        const float d0 = v.x - mean[t], d1 = v.y - mean[t], d2 = v.z - mean[t], d3 = v.w - mean[t];
                        p += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
                    }
                    p = wave_sum(p);
                    if (lane == 0) red[vw*8 + t] = p;
                }
            } else {
                float s = 0.0f;
                for (int e4 = tid; e4 < K4; e4 += nthreads) {
                    const float4 v = *(const float4 *) (stage + (size_t) t*K + e4*4);
                    const float d0 = v.x - mean[t], d1 = v.y - mean[t], d2 = v.z - mean[t], d3 = v.w - mean[t];
                    s += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
                }
                s = wave_sum(s);
                if (lane == 0) red[wave*8 + t] = s;
            }
        }
        __syncthreads();
        #pragma unroll
        for (int t = 0; t < T; t++) {
            float s = 0.0f;
            if (vtree) { for (int w = 0; w < 8; w++) { const float pw = red[(w < nvw ? w : 0)*8 + t]; s += w < nvw ? pw : 0.0f; } }
            else for (int w = 0; w < nwaves; w++) s += red[w*8 + t];
            part[t] = 1.0f / sqrtf(s / K + a.eps);
        }
        for (int idx = tid; idx < T*K4; idx += nthreads) {
            const int t = idx / K4, e4 = idx - t*K4;
            float mt = mean[0], sc = part[0];
            #pragma unroll
            for (int tt = 1; tt < T; tt++) { mt = t == tt ? mean[tt] : mt; sc = t == tt ? part[tt] : sc; }
            const float4 v = *(const float4 *) (stage + (size_t) t*K + e4*4);
            const float4 w = *(const float4 *) (a.ln_w + e4*4);
            const float4 b = *(const float4 *) (a.ln_b + e4*4);
            float o[4] = { (v.x - mt) * sc, (v.y - mt) * sc, (v.z - mt) * sc, (v.w - mt) * sc };
            o[0] = o[0]*w.x; o[1] = o[1]*w.y; o[2] = o[2]*w.z; o[3] = o[3]*w.w;
            o[0] = o[0]+b.x; o[1] = o[1]+b.y; o[2] = o[2]+b.z; o[3] = o[3]+b.w;
            act_store(o, e4*4, t);
        }
    } else if (!a.xfirst) {
        for (int idx = tid; idx < T*K4; idx += nthreads) {
            const int t = idx / K4, e4 = idx - t*K4;
            const float4 x4 = staged ? *(const float4 *) (stage + (size_t) t*K + e4*4)
                                     : *(const float4 *) ((const char *) a.x + (int64_t) t*a.x_nb1 + (int64_t) e4*16);
            const float v[4] = { x4.x, x4.y, x4.z, x4.w };
            act_store(v, e4*4, t);
        }
    }
    __syncthreads();

    // ---- main loop: chunk `it` in registers, chunk it+1 in flight ----
    const uint4 * alo = (const uint4 *) lo;
    const uint4 * ahi = (const uint4 *) hi;
    float acc[T], accm[Q4K ? T : 1];
    #pragma unroll
    for (int t = 0; t < T; t++) acc[t] = 0.0f;
    #pragma unroll
    for (int t = 0; t < (Q4K ? T : 1); t++) accm[t] = 0.0f;

    // two chunks in flight per wave: chunk 0 and chunk 1 were requested BEFORE the prologue, a buffer is refilled with chunk + 2 as soon
    // as it has been used (in-order return: the wait for buffer A leaves buffer B's loads outstanding)
    for (int it0 = 0; it0 < total; it0 += 2) {
      #pragma unroll
      for (int half = 0; half < 2; half++) {
        const int it = it0 + half;
        if (it >= total) break;
        wblk<WT> * buf = half == 0 ? cur : nxt;             // (static after unrolling: no lambda here — capturing acc[] by reference
                                                            //  sends the (j8 == t) select chain below through scratch memory)
        const int pass = it / nchunks, c = it - pass*nchunks;
        #pragma unroll
        for (int u = 0; u < U; u++) {
            const int g = j8 + LPR*(c*U + u);
            if constexpr (Q4K) {
                if (g < nb) wblk_dot_q4k<T>(buf[u], g, 1.0f, nb, nsb, alo, dx, sx, acc, accm);
            } else if (g < nb) {
                uint32_t vlo[4], vhi[4];
                wblk_unpack<WT>(buf[u], vlo, vhi);
                const float dw = h2f(buf[u].d);
                constexpr int off = WT == MI355X_TYPE_Q5_0 ? 16 : (WT == MI355X_TYPE_Q4_0 ? 8 : 0);
                #pragma unroll
                for (int t = 0; t < T; t++) {
                    const uint4 al = alo[(size_t) t*nb + g], ah = ahi[(size_t) t*nb + g];
                    int sum = 0;
                    sum = __builtin_amdgcn_sdot4((int) vlo[0], (int) al.x, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[1], (int) al.y, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[2], (int) al.z, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[3], (int) al.w, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[0], (int) ah.x, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[1], (int) ah.y, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[2], (int) ah.z, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[3], (int) ah.w, sum, false);
                    if (off) sum -= off * sx[t*nb + g];
                    acc[t] = fmaf(dw * dx[t*nb + g], (float) sum, acc[t]);
                }
            }
        }
        if (c == nchunks - 1) {
            // reduce over the LPR lanes of the row, lane j8 == t finishes column t
            #pragma unroll
            for (int t = 0; t < T; t++) {
                if constexpr (Q4K) { acc[t] += accm[t]; accm[t] = 0.0f; }
                acc[t] = group_sum<LPR>(acc[t]);
            }
            float v = acc[0];
            #pragma unroll
            for (int t = 1; t < T; t++) v = (j8 == t) ? acc[t] : v;
            int s, row; const bool rok = row_of(pass, s, row);
            if (rok && j8 < T) {
                const DGSeg & sg = a.seg[s];
                if (sg.bias)      v = v + sg.bias[row];
                if (sg.has_scale) v = v * sg.scale;
                if (sg.gelu)      v = gelu_lut(v, a.gelu_tab);
                if (sg.residual)  v = v + *(const float *) ((const char *) sg.residual + (int64_t) j8*sg.res_nb1 + (int64_t) row*4);
                char * dp = (char *) sg.dst + (int64_t) j8*sg.dst_nb1;
                if (a.use_cols) {
                    dp = (char *) a.dstcol[0];
                    #pragma unroll
                    for (int t = 1; t < T; t++) dp = (j8 == t) ? (char *) a.dstcol[t] : dp;
                }
                if (sg.dst_f16) ((uint16_t *) dp)[row] = f2h(v); else ((float *) dp)[row] = v;
                if (a.mircol[0]) {
                    char * mp = (char *) a.mircol[0];
                    #pragma unroll
                    for (int t = 1; t < T; t++) mp = (j8 == t) ? (char *) a.mircol[t] : mp;
                    ((float *) mp)[row] = v;
                }
            }
            #pragma unroll
            for (int t = 0; t < T; t++) acc[t] = 0.0f;
        }
        if (it + 2 < total) load_chunk(buf, it + 2);
      }
    }
}

template <int WT, int LPR>
static int launch_gemv8_T(mi355x_ctx * ctx, const DGArgs & k, int T, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    const char * name = "gemv";
    switch (T) {
        case 1: return emit(ctx, name, k_gemv8<WT, 1, LPR>, grid, block, lds, k, bytes, flops);
        case 2: return emit(ctx, name, k_gemv8<WT, 2, LPR>, grid, block, lds, k, bytes, flops);
        case 3: return emit(ctx, name, k_gemv8<WT, 3, LPR>, grid, block, lds, k, bytes, flops);
        case 4: return emit(ctx, name, k_gemv8<WT, 4, LPR>, grid, block, lds, k, bytes, flops);
        case 5: return emit(ctx, name, k_gemv8<WT, 5, LPR>, grid, block, lds, k, bytes, flops);
        case 6: return emit(ctx, name, k_gemv8<WT, 6, LPR>, grid, block, lds, k, bytes, flops);
        case 7: return emit(ctx, name, k_gemv8<WT, 7, LPR>, grid, block, lds, k, bytes, flops);
        case 8: return emit(ctx, name, k_gemv8<WT, 8, LPR>, grid, block, lds, k, bytes, flops);
        default: return MI355X_E_UNSUPPORTED;
    }
}
template <int WT>
static int launch_gemv8(mi355x_ctx * ctx, const DGArgs & k, int T, int lpr, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    switch (lpr) {
        case 8:  return launch_gemv8_T<WT, 8>(ctx, k, T, grid, block, lds, bytes, flops);
        case 16: return launch_gemv8_T<WT, 16>(ctx, k, T, grid, block, lds, bytes, flops);
        case 32: return launch_gemv8_T<WT, 32>(ctx, k, T, grid, block, lds, bytes, flops);
        default: return launch_gemv8_T<WT, 64>(ctx, k, T, grid, block, lds, bytes, flops);
    }
}


// ---------------------------------------------------------------------------------------------------
// k_gemv_row: the lean mat-vec for the per-layer projections (N <= 8192 rows): ONE weight row per wave, 4 rows per
// workgroup, so a 1280-row matrix puts 1280 waves' loads in flight right after launch.  Everything that depends on the
// row is wave-uniform (scalar); activations are held in registers from load to quantization (no staging, no integer
// division, three barriers with LayerNorm, one without); all cross-lane traffic is DPP / permlane.
//
// EVERY global load of the kernel is issued in one straight-line burst at the top, in the order short-latency first
// (activations / attention partials: L2 or Infinity-Cache hits) ... weights last (HBM), with clamped always-valid
// addresses instead of predicates.  Two compiler facts force this shape (ROCm 7.2): a load inside a conditional whose
// result merges with another value (phi) is waited for at the end of its basic block, i.e. `ok ? *p : 0` serializes
// every such load behind an `s_waitcnt vmcnt(0)`; and loads return in issue order, so anything requested after the
// weights cannot be consumed before they arrive.  MODE selects the activation source at compile time for the same
// reason (no phi between the sources): 0 plain x, 1 LayerNorm(x)*w+b, 2 combine of attention partial records.
// ---------------------------------------------------------------------------------------------------
// R = weight rows per wave.  1 for T <= 2.  With more columns every workgroup re-reads T*K*4 bytes of activations from L2
// (26 MB per launch for 1024 workgroups at T = 5, six times the weight bytes): R = 4 rows per wave cuts the workgroup count
// and that traffic by 4 while the same number of weight loads stays in flight.
template <int WT, int T, int XS, int MODE, bool NSEG1, int R = 1>                   // XS = float4 activation slots per column per thread: 1 (K <= 2048, threads >= K/4) or 5 (K <= 5120, 256 threads)
__global__ void __launch_bounds__(512) k_gemv_row(const DGArgs a) {
    constexpr bool Q4K = WT == MI355X_TYPE_Q4_K;
    constexpr int NU = Q4K ? (XS == 1 ? 1 : 2) : (XS == 1 ? 1 : 3);      // units per lane: 32-element blocks (64-element chunks for Q4_K)
    constexpr int MAXP = 12;                                             // attention partial records per (head, query) handled in registers
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 4..8 waves per workgroup, one row each; the launcher picks ceil(K/256) waves when that lets a single activation slot
    // per thread cover the whole vector (K = 1280 -> 5 waves = 320 threads)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int K = a.K, nb = Q4K ? K >> 6 : K >> 5, K4 = K >> 2;           // nb = lane-units per row
    const int nsb = K >> 8;
    const int ntot = a.ntot;
    const int grow = __builtin_amdgcn_readfirstlane((blockIdx.x * nwaves + wave) * R);      // first of this wave's R rows (same segment, all valid or none: launcher)
    int s = 0;
    if constexpr (!NSEG1) {
        if (a.nseg > 1 && grow >= a.row_start[1]) s = 1;
        if (a.nseg > 2 && grow >= a.row_start[2]) s = 2;
    }
    const bool rok = grow < ntot;
    const int row = rok ? grow - (NSEG1 ? 0 : a.row_start[s]) : 0;       // clamped: always a valid row of segment s
    // Single-segment launches (NSEG1: everything but Q/K/V) read the segment at a static kernarg offset, so ALL scalar
    // loads of the kernel form one batch; with several segments the (wave-uniform) segment index costs one more dependent
    // scalar round trip.  Every field — the epilogue's parameters included — is pinned into SGPRs HERE: left to the
    // compiler they are fetched one by one at their first use, i.e. serialized scalar-cache misses at the kernel tail.
    const DGSeg & sgr = NSEG1 ? a.seg[0] : a.seg[s];
    struct { const void * w; int64_t nbt; const float * bias; const float * residual; int64_t res_nb1; void * dst; int64_t dst_nb1;
             float scale; int has_scale, gelu, dst_f16; } sg;
    sg.w = sgr.w; sg.nbt = sgr.nbt; sg.bias = sgr.bias; sg.residual = sgr.residual; sg.res_nb1 = sgr.res_nb1;
    sg.dst = sgr.dst; sg.dst_nb1 = sgr.dst_nb1; sg.scale = sgr.scale; sg.has_scale = sgr.has_scale; sg.gelu = sgr.gelu; sg.dst_f16 = sgr.dst_f16;
    asm volatile("" :: "s"(sg.w), "s"(sg.nbt), "s"(sg.bias), "s"(sg.residual), "s"(sg.res_nb1), "s"(sg.dst), "s"(sg.dst_nb1),
                       "s"(sg.scale), "s"(sg.has_scale), "s"(sg.gelu), "s"(sg.dst_f16), "s"(a.gelu_tab),
                       "s"(a.x), "s"(a.x_nb1), "s"(a.ln_w), "s"(a.ln_b), "s"(a.part_o), "s"(a.part_ml), "s"(a.nparts), "s"(a.eps));
    DG_STAMP(0);

    // ---- the load burst -------------------------------------------------------------------------------------------
    float4 xr[T][XS];
    constexpr bool PBURST = MODE == 2 && T <= 2;                         // record bursts cost 72 registers per column
    float2 pml[PBURST ? MAXP : 1];
    float4 pov[PBURST ? MAXP : 1];
    const int pe4 = tid < K4 ? tid : K4 - 1, ph = pe4 >> 4, pd = (pe4 & 15) << 2;      // MODE 2: this thread's slot-0 element
    if constexpr (MODE == 2 && !PBURST) {
        // T > 2 (beam search): plain two-pass combine per column after the weights have been requested (below)
    } else if constexpr (MODE == 2) {
        // slot 0 of column 0: all records at once (the other columns / slots follow after the weights are requested)
        const int64_t base = ((int64_t) ph*T + 0) * a.nparts;
        #pragma unroll
        for (int p = 0; p < MAXP; p++) {
            const int pc = p < a.nparts ? p : a.nparts - 1;
            pml[p] = *(const float2 *) (a.part_ml + (base + pc)*2);
            pov[p] = *(const float4 *) (a.part_o + (base + pc)*64 + pd);
        }
    } else {
        #pragma unroll
        for (int t = 0; t < T; t++) {
            const char * xp = (const char *) a.x + (int64_t) t*a.x_nb1;
            #pragma unroll
            for (int i = 0; i < XS; i++) {
                const int e4 = tid + i*nthreads;
                xr[t][i] = *(const float4 *) (xp + (size_t) (e4 < K4 ? e4 : K4 - 1)*16);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // pin the issue order: the scheduler otherwise sinks / reorders these loads
    float4 lw[MODE == 1 ? XS : 1], lb[MODE == 1 ? XS : 1];
    if constexpr (MODE == 1) {
        #pragma unroll
        for (int i = 0; i < XS; i++) {
            const int e4 = tid + i*nthreads, e4c = e4 < K4 ? e4 : K4 - 1;
            lw[i] = *(const float4 *) (a.ln_w + e4c*4);
            lb[i] = *(const float4 *) (a.ln_b + e4c*4);
        }
    }
    // bias / residual of (row, column lane): clamped column, dummy (valid) address when absent
    // lane r*T + t finishes (row r, column t)
    const int tcol = R == 1 ? (lane < T ? lane : T - 1) : lane % T;
    const int rlane = R == 1 ? 0 : (lane / T < R ? lane / T : R - 1);
    const float * bptr = sg.bias ? sg.bias + row + rlane : (const float *) a.gelu_tab;
    const float * rptr = sg.residual ? (const float *) ((const char *) sg.residual + (int64_t) tcol*sg.res_nb1) + row + rlane : (const float *) a.gelu_tab;
    const float bias_v = *bptr, res_v = *rptr;
    __builtin_amdgcn_sched_barrier(0);
    wblk<WT> wr[R][NU];
    {
        const char * base = (const char *) sg.w;
        const int64_t nbt = sg.nbt;
        #pragma unroll
        for (int r = 0; r < R; r++) {
            const int ib0 = (row + r) * nb;
            #pragma unroll
            for (int u = 0; u < NU; u++) {
                const int g = lane + 64*u, gc = g < nb ? g : nb - 1;
                if constexpr (Q4K) wblk_load_q4k(wr[r][u], base, nbt, row + r, nsb, gc);
                else               wblk_load<WT>(wr[r][u], base, nbt, (int64_t) (ib0 + gc));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    DG_STAMP(1);                                                        // all loads issued

    float * red = (float *) smem;                                       // [2][T][8]
    uint32_t * lo = (uint32_t *) (smem + 512);                           // Q4_K: the four Q8_K planes [4][T][K/64] uint4
    uint32_t * hi = lo + (size_t) T*nb*4;
    float * dx = Q4K ? (float *) (smem + 512 + (size_t) T*K) : (float *) (hi + (size_t) T*nb*4);
    int *   sx = Q4K ? (int *) (dx + T*nsb) : (int *) (dx + T*nb);       // Q4_K: per-32-element sums
    auto act_store = [&](const float v[4], int e, int t) {
        if constexpr (Q4K) dg_q8_K_store(v, e, t, K, T, lo, dx, sx);
        else               dg_q8_0_store(v, e, t, nb, lo, hi, dx, sx);
    };

    // ---- activations -> registers ---------------------------------------------------------------------------------
    float4 * xstage = (float4 *) (smem + ((((char *) sx - smem) + (Q4K ? (size_t) T*(K >> 5)*4 : (size_t) T*nb*4) + 15) & ~(size_t) 15));   // MODE 2, T > 2 only
    if constexpr (MODE == 2 && !PBURST) {
        // one record burst per column in a ROLLED loop (unrolled, the compiler hoists all columns' loads: 72 registers
        // each); the combined column goes through LDS because a rolled loop cannot index the register array
        #pragma unroll 1
        for (int t = 0; t < T; t++) {
            float2 ml[MAXP]; float4 ov[MAXP];
            const int64_t base = ((int64_t) ph*T + t) * a.nparts;
            #pragma unroll
            for (int p = 0; p < MAXP; p++) {
                const int pc = p < a.nparts ? p : a.nparts - 1;
                ml[p] = *(const float2 *) (a.part_ml + (base + pc)*2);
                ov[p] = *(const float4 *) (a.part_o + (base + pc)*64 + pd);
            }
            float M = -1e30f, L = 0.0f;
            float4 o = make_float4(0, 0, 0, 0);
            #pragma unroll
            for (int p = 0; p < MAXP; p++) M = fmaxf(M, ml[p].x);
            #pragma unroll
            for (int p = 0; p < MAXP; p++) {
                const float w = p < a.nparts ? __expf(ml[p].x - M) : 0.0f;
                L = fmaf(w, ml[p].y, L);
                o.x = fmaf(w, ov[p].x, o.x); o.y = fmaf(w, ov[p].y, o.y); o.z = fmaf(w, ov[p].z, o.z); o.w = fmaf(w, ov[p].w, o.w);
            }
            const float inv = L == 0.0f ? 0.0f : 1.0f / L;
            xstage[t*nthreads + tid] = make_float4(o.x*inv, o.y*inv, o.z*inv, o.w*inv);
        }
    }
    if constexpr (MODE == 2) {
        // x[t][h*64 + d] = sum_p w_p o_p[d] / sum_p w_p l_p,  w_p = exp(m_p - max_p m_p)   (k_fattn_dec records)
        #pragma unroll
        for (int t = 0; t < T; t++) {
            float M = -1e30f, L = 0.0f;
            float4 o = make_float4(0, 0, 0, 0);
            if constexpr (PBURST) {
                if (t > 0) {                                                    // one burst per further column
                    const int64_t base = ((int64_t) ph*T + t) * a.nparts;
                    #pragma unroll
                    for (int p = 0; p < MAXP; p++) {
                        const int pc = p < a.nparts ? p : a.nparts - 1;
                        pml[p] = *(const float2 *) (a.part_ml + (base + pc)*2);
                        pov[p] = *(const float4 *) (a.part_o + (base + pc)*64 + pd);
                    }
                }
                #pragma unroll
                for (int p = 0; p < MAXP; p++) M = fmaxf(M, pml[p].x);          // duplicates of the last record do not change the max
                #pragma unroll
                for (int p = 0; p < MAXP; p++) {
                    const float w = p < a.nparts ? __expf(pml[p].x - M) : 0.0f;
                    L = fmaf(w, pml[p].y, L);
                    o.x = fmaf(w, pov[p].x, o.x); o.y = fmaf(w, pov[p].y, o.y); o.z = fmaf(w, pov[p].z, o.z); o.w = fmaf(w, pov[p].w, o.w);
                }
            } else {
                // T > 2: the column's combined value was staged in LDS by the rolled loop above
                const float4 sv = xstage[t*nthreads + tid];
                xr[t][0] = sv;
                continue;
            }
            const float inv = L == 0.0f ? 0.0f : 1.0f / L;
            xr[t][0] = make_float4(o.x*inv, o.y*inv, o.z*inv, o.w*inv);
            #pragma unroll
            for (int i = 1; i < XS; i++) {
                const int e4 = tid + i*nthreads;
                float4 o2 = make_float4(0, 0, 0, 0);
                if (e4 < K4) {
                    const int h = e4 >> 4, d = (e4 & 15) << 2;
                    const int64_t base = ((int64_t) h*T + t) * a.nparts;
                    float M2 = -1e30f, L2 = 0.0f;
                    for (int p = 0; p < a.nparts; p++) M2 = fmaxf(M2, a.part_ml[(base + p)*2]);
                    for (int p = 0; p < a.nparts; p++) {
                        const float2 ml = *(const float2 *) (a.part_ml + (base + p)*2);
                        const float w = __expf(ml.x - M2);
                        const float4 v = *(const float4 *) (a.part_o + (base + p)*64 + d);
                        L2 = fmaf(w, ml.y, L2);
                        o2.x = fmaf(w, v.x, o2.x); o2.y = fmaf(w, v.y, o2.y); o2.z = fmaf(w, v.z, o2.z); o2.w = fmaf(w, v.w, o2.w);
                    }
                    const float inv2 = L2 == 0.0f ? 0.0f : 1.0f / L2;
                    o2 = make_float4(o2.x*inv2, o2.y*inv2, o2.z*inv2, o2.w*inv2);
                }
                xr[t][i] = o2;
            }
        }
    }
    if constexpr (MODE == 1) {
        // ggml_norm (ggml-cpu/ops.cpp:3698-3765) + affine, two passes over the registers
        float mean[T], rstd[T];
        #pragma unroll
        for (int t = 0; t < T; t++) {
            float p = 0.0f;
            #pragma unroll
            for (int i = 0; i < XS; i++) if (tid + i*nthreads < K4) p += (xr[t][i].x + xr[t][i].y) + (xr[t][i].z + xr[t][i].w);
            p = wave_sum(p);
            if (lane == 0) red[t*8 + wave] = p;
        }
        DG_STAMP(2);                                                    // activations arrived, first reduction done
        __syncthreads();
        #pragma unroll
        for (int t = 0; t < T; t++) {
            // the (up to 8) wave partials: eight independent LDS reads and a value mask instead of a rolled loop of dependent
            // read-add steps (nwaves is a launch parameter); same summation order, slots >= nwaves hold stale data
            float pw[8];
            #pragma unroll
            for (int w = 0; w < 8; w++) pw[w] = red[t*8 + w];
            float rs = 0.0f;
            #pragma unroll
            for (int w = 0; w < 8; w++) rs += w < nwaves ? pw[w] : 0.0f;
            mean[t] = rs / K;
            float p = 0.0f;
            #pragma unroll
            for (int i = 0; i < XS; i++) {
                if (tid + i*nthreads < K4) {
                    const float d0 = xr[t][i].x - mean[t], d1 = xr[t][i].y - mean[t], d2 = xr[t][i].z - mean[t], d3 = xr[t][i].w - mean[t];
                    p += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
                }
            }
            p = wave_sum(p);
            if (lane == 0) red[T*8 + t*8 + wave] = p;
        }
        __syncthreads();
        #pragma unroll
        for (int t = 0; t < T; t++) {
            float pw[8];
            #pragma unroll
            for (int w = 0; w < 8; w++) pw[w] = red[T*8 + t*8 + w];
            float rs = 0.0f;
            #pragma unroll
            for (int w = 0; w < 8; w++) rs += w < nwaves ? pw[w] : 0.0f;
            rstd[t] = 1.0f / sqrtf(rs / K + a.eps);
            #pragma unroll
            for (int i = 0; i < XS; i++) {
                const int e4 = tid + i*nthreads;
                if (e4 < K4) {
                    const float4 w = lw[i], b = lb[i];
                    float o[4] = { (xr[t][i].x - mean[t]) * rstd[t], (xr[t][i].y - mean[t]) * rstd[t], (xr[t][i].z - mean[t]) * rstd[t], (xr[t][i].w - mean[t]) * rstd[t] };
                    o[0] = o[0]*w.x; o[1] = o[1]*w.y; o[2] = o[2]*w.z; o[3] = o[3]*w.w;
                    o[0] = o[0]+b.x; o[1] = o[1]+b.y; o[2] = o[2]+b.z; o[3] = o[3]+b.w;
                    act_store(o, e4*4, t);
                }
            }
        }
    } else {
        #pragma unroll
        for (int t = 0; t < T; t++) {
            #pragma unroll
            for (int i = 0; i < XS; i++) {
                const int e4 = tid + i*nthreads;
                if (e4 < K4) {
                    const float v[4] = { xr[t][i].x, xr[t][i].y, xr[t][i].z, xr[t][i].w };
                    act_store(v, e4*4, t);
                }
            }
        }
    }
    DG_STAMP(3);                                                        // activations quantized into LDS
    __syncthreads();
    DG_STAMP(4);

    // ---- dot products: lane handles blocks lane (+64, +128) of the wave's row; clamped duplicates carry weight 0 ----
    const uint4 * alo = (const uint4 *) lo;
    const uint4 * ahi = (const uint4 *) hi;
    float acc[R][T];
    #pragma unroll
    for (int r = 0; r < R; r++)
        #pragma unroll
        for (int t = 0; t < T; t++) acc[r][t] = 0.0f;
    if constexpr (Q4K) {
        #pragma unroll
        for (int r = 0; r < R; r++) {
            float accm[T];
            #pragma unroll
            for (int t = 0; t < T; t++) accm[t] = 0.0f;
            #pragma unroll
            for (int u = 0; u < NU; u++) {
                const int g = lane + 64*u, gc = g < nb ? g : nb - 1;
                wblk_dot_q4k<T>(wr[r][u], gc, g < nb ? 1.0f : 0.0f, nb, nsb, alo, dx, sx, acc[r], accm);
            }
            #pragma unroll
            for (int t = 0; t < T; t++) acc[r][t] += accm[t];
        }
    } else {
        #pragma unroll
        for (int u = 0; u < NU; u++) {
            const int g = lane + 64*u, gc = g < nb ? g : nb - 1;
            uint32_t vlo[R][4], vhi[R][4];
            float dw[R];
            #pragma unroll
            for (int r = 0; r < R; r++) {
                wblk_unpack<WT>(wr[r][u], vlo[r], vhi[r]);
                dw[r] = g < nb ? h2f(wr[r][u].d) : 0.0f;
            }
            constexpr int off = WT == MI355X_TYPE_Q5_0 ? 16 : (WT == MI355X_TYPE_Q4_0 ? 8 : 0);
            #pragma unroll
            for (int t = 0; t < T; t++) {
                const uint4 al = alo[(size_t) t*nb + gc], ah = ahi[(size_t) t*nb + gc];       // one LDS read serves all R rows
                const int sxv = off ? sx[t*nb + gc] : 0;
                const float dxv = dx[t*nb + gc];
                #pragma unroll
                for (int r = 0; r < R; r++) {
                    int sum = 0;
                    sum = __builtin_amdgcn_sdot4((int) vlo[r][0], (int) al.x, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[r][1], (int) al.y, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[r][2], (int) al.z, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[r][3], (int) al.w, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[r][0], (int) ah.x, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[r][1], (int) ah.y, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[r][2], (int) ah.z, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[r][3], (int) ah.w, sum, false);
                    if (off) sum -= off * sxv;
                    acc[r][t] = fmaf(dw[r] * dxv, (float) sum, acc[r][t]);
                }
            }
        }
    }
    DG_STAMP(5);                                                        // weights arrived, dots done
    #pragma unroll
    for (int r = 0; r < R; r++)
        #pragma unroll
        for (int t = 0; t < T; t++) acc[r][t] = wave_sum(acc[r][t]);
    float v = acc[0][0];
    #pragma unroll
    for (int r = 0; r < R; r++)
        #pragma unroll
        for (int t = 0; t < T; t++) if (r + t > 0) v = (lane == r*T + t) ? acc[r][t] : v;
    if (rok && lane < R*T) {
        if (sg.bias)      v = v + bias_v;
        if (sg.has_scale) v = v * sg.scale;
        if (sg.gelu)      v = gelu_lut(v, a.gelu_tab);
        if (sg.residual)  v = v + res_v;
        char * dp = (char *) sg.dst + (int64_t) tcol*sg.dst_nb1;
        if (sg.dst_f16) ((uint16_t *) dp)[row + rlane] = f2h(v); else ((float *) dp)[row + rlane] = v;
    }
    DG_STAMP(6);
}

template <int WT, int MODE, bool NSEG1>
static int launch_gemv_row_m(mi355x_ctx * ctx, const DGArgs & k, int T, dim3 grid, uint32_t lds, double bytes, double flops, int R = 1) {
    const char * name = "gemv";
    if constexpr (MODE == 0 || MODE == 1) {
        if (R == 4 && k.K <= 2048) {            // four rows per wave (grid already sized for it by the caller)
            const dim3 block4(64 * gemv_row_waves(k.K));
            switch (T) {
                case 1: return emit(ctx, name, k_gemv_row<WT, 1, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                case 2: return emit(ctx, name, k_gemv_row<WT, 2, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                case 3: return emit(ctx, name, k_gemv_row<WT, 3, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                case 4: return emit(ctx, name, k_gemv_row<WT, 4, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                case 5: return emit(ctx, name, k_gemv_row<WT, 5, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                case 6: return emit(ctx, name, k_gemv_row<WT, 6, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                case 7: return emit(ctx, name, k_gemv_row<WT, 7, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                case 8: return emit(ctx, name, k_gemv_row<WT, 8, 1, MODE, NSEG1, 4>, grid, block4, lds, k, bytes, flops);
                default: return MI355X_E_UNSUPPORTED;
            }
        }
    }
    if (k.K > 2048) {
        if constexpr (MODE == 0 || MODE == 1) { if (T == 1) return emit(ctx, name, k_gemv_row<WT, 1, 5, MODE, NSEG1>, grid, dim3(256), lds, k, bytes, flops); }
        return MI355X_E_UNSUPPORTED;
    }
    const dim3 block(64 * gemv_row_waves(k.K));
    switch (T) {
        case 1: return emit(ctx, name, k_gemv_row<WT, 1, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        case 2: return emit(ctx, name, k_gemv_row<WT, 2, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        case 3: return emit(ctx, name, k_gemv_row<WT, 3, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        case 4: return emit(ctx, name, k_gemv_row<WT, 4, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        case 5: return emit(ctx, name, k_gemv_row<WT, 5, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        case 6: return emit(ctx, name, k_gemv_row<WT, 6, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        case 7: return emit(ctx, name, k_gemv_row<WT, 7, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        case 8: return emit(ctx, name, k_gemv_row<WT, 8, 1, MODE, NSEG1>, grid, block, lds, k, bytes, flops);
        default: return MI355X_E_UNSUPPORTED;
    }
}
template <int WT>
static int launch_gemv_row(mi355x_ctx * ctx, const DGArgs & k, int T, dim3 grid, uint32_t lds, double bytes, double flops, int R = 1) {
    if (k.x == nullptr) {
        if (k.nparts > 12 || k.nseg != 1) return MI355X_E_UNSUPPORTED;   // records of one column are held in registers, 12 at most
        return launch_gemv_row_m<WT, 2, true>(ctx, k, T, grid, lds, bytes, flops, R);
    }
    if (k.has_norm) {
        return k.nseg == 1 ? launch_gemv_row_m<WT, 1, true>(ctx, k, T, grid, lds, bytes, flops, R) : launch_gemv_row_m<WT, 1, false>(ctx, k, T, grid, lds, bytes, flops, R);
    }
    if (k.nseg != 1) return MI355X_E_UNSUPPORTED;                        // several segments without LayerNorm: generic kernel
    return launch_gemv_row_m<WT, 0, true>(ctx, k, T, grid, lds, bytes, flops, R);
}

// second-generation entry: returns MI355X_E_UNSUPPORTED for anything it does not cover (caller falls back to k_gemv)
int mi355x_gemv8(mi355x_ctx * ctx, const mi355x_gemv_desc * d) {
    if (d->nseg < 1 || d->nseg > 3 || d->T < 1 || d->T > 8) return MI355X_E_UNSUPPORTED;
    const int wt = d->seg[0].wtype, K = d->K, T = d->T;
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0 && wt != MI355X_TYPE_Q4_K) return MI355X_E_UNSUPPORTED;
    if (K <= 0 || K % 32 || (wt == MI355X_TYPE_Q4_K && K % 256)) return MI355X_E_UNSUPPORTED;
    const bool from_part = d->attn_part_o != nullptr;
    const bool from_q = d->x_planes != nullptr;                          // planes made by decode_q.hip: only k_gemv8 below takes them here
    if (from_q) {
        if (d->x || from_part || d->has_norm || d->planes_out || ((uintptr_t) d->x_planes % 16)) return MI355X_E_UNSUPPORTED;
        if (d->cols && d->nseg != 1) return MI355X_E_UNSUPPORTED;
    } else if (from_part) {
        if (d->x || !d->attn_part_ml || d->attn_nparts < 1 || d->attn_nparts > 64 || K % 64 || d->has_norm) return MI355X_E_UNSUPPORTED;
    } else if (!d->x || ((uintptr_t) d->x % 16) || (d->x_nb1 % 16)) return MI355X_E_UNSUPPORTED;
    if (d->has_norm && (!d->ln_w || !d->ln_b || ((uintptr_t) d->ln_w % 16) || ((uintptr_t) d->ln_b % 16))) return MI355X_E_UNSUPPORTED;
    const bool staged = from_part || d->has_norm;
    const size_t lds = dg_lds_bytes(wt, K, T, staged);

    DGArgs k; memset(&k, 0, sizeof(k));
    k.x = from_part ? nullptr : d->x; k.x_nb1 = d->x_nb1; k.K = K; k.has_norm = d->has_norm; k.eps = d->eps; k.nseg = d->nseg;
    k.ln_w = d->ln_w; k.ln_b = d->ln_b; k.gelu_tab = ctx->gelu_tab;
    k.dbg = (unsigned long long *) mi355x_debug_stamps(ctx);
    k.part_o = d->attn_part_o; k.part_ml = d->attn_part_ml; k.nparts = d->attn_nparts;
    k.xq = d->x_planes;
    ctx->last_mirrored = 0;
    const bool want_mirror = d->cols && d->cols->mirror[0] && d->nseg == 1 && d->seg[0].dst_type == MI355X_TYPE_F32;
    if (want_mirror) for (int t = 0; t < 8; t++) { k.mircol[t] = d->cols->mirror[t < T ? t : T - 1]; if (!k.mircol[t]) return MI355X_E_UNSUPPORTED; }
    if (from_q && d->cols) {
        k.use_cols = 1;
        for (int t = 0; t < 8; t++) k.dstcol[t] = d->cols->dst[0][t < T ? t : T - 1];
        for (int t = 0; t < T; t++) if (!k.dstcol[t] || d->cols->res[0][t]) return MI355X_E_UNSUPPORTED;
    }
    int ntot = 0; double wbytes = 0;
    for (int s = 0; s < d->nseg; s++) {
        const mi355x_gemv_seg & g = d->seg[s];
        if (g.wtype != wt || g.N <= 0 || ((uintptr_t) g.w % 16)) return MI355X_E_UNSUPPORTED;
        if (g.dst_type != MI355X_TYPE_F32 && g.dst_type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
        k.row_start[s] = ntot;
        DGSeg & o = k.seg[s];
        o.w = g.w; o.N = g.N; o.nbt = wt == MI355X_TYPE_Q4_K ? (int64_t) g.N * (K / 256) : (int64_t) g.N * (K / 32);
        o.bias = g.ep.bias; o.scale = g.ep.scale; o.has_scale = g.ep.has_scale; o.gelu = g.ep.gelu;
        o.residual = g.ep.residual; o.res_nb1 = g.ep.residual_nb1;
        o.dst = g.dst; o.dst_nb1 = g.dst_nb1; o.dst_f16 = g.dst_type == MI355X_TYPE_F16;
        ntot += g.N;
        wbytes += (double) mi355x_type_row_bytes(wt, K) * g.N;
    }
    for (int s = d->nseg; s < 4; s++) k.row_start[s] = ntot;
    k.ntot = ntot;
    const double bytes0 = wbytes + (double) K*T*4 + (double) ntot*T*4;
    const double flops0 = 2.0 * ntot * K * T;
    if (!from_q && !want_mirror && ntot <= 8192 && K <= (T == 1 ? 5120 : 2048)) {
        const int rpw = gemv_row_waves(K);
        // 512-byte reduction header + activation planes (+ one float4 per thread and column when the attention combine
        // of T > 2 columns is staged through LDS)
        const size_t lds_row = 512 + dg_act_bytes(wt, K, T) + ((from_part && T > 2) ? (size_t) T * 64 * rpw * 16 + 16 : 0);
        // rows per wave: 4 once the activation re-reads outweigh the weights (T >= 3), if every segment is a multiple of 4 rows
        // (measured at T = 5, large-v3: LN+FC1 11.3 -> 10.0 us, LN+QKV 12.5 -> 10.5 us; the 1280-row O-projection with its
        // attention-combine prologue gets SLOWER on 64 workgroups, 12.1 -> 16.4 us, so it keeps one row per wave)
        // Also at T = 1 / 2 for the wide mat-vecs (LN + Q/K/V, LN + fc1): fewer, fatter workgroups stage the activation fewer times
        // (round-3 A-B, large-v3 Q5_0: 1.391 -> 1.330 ms/token, profiles/r03_ab_gemv_rows_min_t.txt)
        int R = (K <= 2048 && !from_part && ntot / (gemv_row_waves(K) * 4) >= 128) ? 4 : 1;
        for (int s = 0; s < d->nseg; s++) if (d->seg[s].N % 4) R = 1;
        const int rpb = gemv_row_waves(K) * R;
        const dim3 grid((ntot + rpb - 1) / rpb);
        int rc = MI355X_E_UNSUPPORTED;
        switch (wt) {
            case MI355X_TYPE_Q4_0: rc = launch_gemv_row<MI355X_TYPE_Q4_0>(ctx, k, T, grid, (uint32_t) lds_row, bytes0, flops0, R); break;
            case MI355X_TYPE_Q5_0: rc = launch_gemv_row<MI355X_TYPE_Q5_0>(ctx, k, T, grid, (uint32_t) lds_row, bytes0, flops0, R); break;
            case MI355X_TYPE_Q8_0: rc = launch_gemv_row<MI355X_TYPE_Q8_0>(ctx, k, T, grid, (uint32_t) lds_row, bytes0, flops0, R); break;
            case MI355X_TYPE_Q4_K: rc = launch_gemv_row<MI355X_TYPE_Q4_K>(ctx, k, T, grid, (uint32_t) lds_row, bytes0, flops0, R); break;
        }
        if (rc != MI355X_E_UNSUPPORTED) return rc;
    }
    if (lds > 64*1024) return MI355X_E_UNSUPPORTED;
    if (wt != MI355X_TYPE_Q4_K && staged && ((size_t) T * (K/32) * 40) % 16) return MI355X_E_UNSUPPORTED;      // float4 alignment of the staging area
    // geometry: latency-bound regime => as many waves as the matrix allows, up to ~16 per CU: lanes per row LPR such that
    // N * LPR / 64 waves >= 8 per CU (but no more lanes than the row has blocks), 4 waves per workgroup (each workgroup
    // repeats the activation prologue; 256 threads keep it short), several passes per wave only for huge N
    constexpr int env_wpc = 8;
    int lpr = 8;
    while (lpr < 64 && (int64_t) ntot * lpr / 64 < (int64_t) ctx->n_cu * env_wpc) lpr *= 2;
    while (lpr > 8 && lpr / 2 >= K / (wt == MI355X_TYPE_Q4_K ? 64 : 32)) lpr /= 2;
    const int rpw = 64 / lpr;
    const int ngroups = (ntot + rpw - 1) / rpw;
    // row groups per wave: one wave per row group up to 32 waves per CU (the whole matrix requested at once), more only beyond that.
    // at most 8 waves per CU, each walking several row groups with the next group's loads in flight (A-B in round 3
    // for the vocabulary projection, 6484 row groups: 32 -> 1 pass, 16 -> 2, 8 -> 4)
    // (measured r03, single stream large-v3 Q5_0: 32 -> 1.3921, 16 -> 1.3888, 8 -> 1.3806 ms/token; profiles/r03_single_stream_mirror_and_logits_passes_ab.txt)
    constexpr int pw = 8;
    int passes = (ngroups + ctx->n_cu*pw - 1) / (ctx->n_cu*pw);
    if (passes < 1) passes = 1; if (passes > 16) passes = 16;
    const int nw = (ngroups + passes - 1) / passes;
    constexpr int wpb = 4;
    k.passes = passes;
    k.xfirst = (!from_part && !from_q && (int64_t) T * (K/4) <= (int64_t) DG_XR * 64 * wpb) ? 1 : 0;
    const int nblocks = (nw + wpb - 1) / wpb;
    const double bytes = wbytes + (double) K*T*4 + (double) ntot*T*4;
    const double flops = 2.0 * ntot * K * T;
    const dim3 grid(nblocks), block(64*wpb);
    int rc8 = MI355X_E_UNSUPPORTED;
    switch (wt) {
        case MI355X_TYPE_Q4_0: rc8 = launch_gemv8<MI355X_TYPE_Q4_0>(ctx, k, T, lpr, grid, block, (uint32_t) lds, bytes, flops); break;
        case MI355X_TYPE_Q5_0: rc8 = launch_gemv8<MI355X_TYPE_Q5_0>(ctx, k, T, lpr, grid, block, (uint32_t) lds, bytes, flops); break;
        case MI355X_TYPE_Q8_0: rc8 = launch_gemv8<MI355X_TYPE_Q8_0>(ctx, k, T, lpr, grid, block, (uint32_t) lds, bytes, flops); break;
        case MI355X_TYPE_Q4_K: rc8 = launch_gemv8<MI355X_TYPE_Q4_K>(ctx, k, T, lpr, grid, block, (uint32_t) lds, bytes, flops); break;
    }
    if (rc8 == 0 && want_mirror) ctx->last_mirrored = 1;
    return rc8;
}

// ---------------------------------------------------------------------------------------------------
// decode attention: partial records per (head, query, 128-key chunk)
//   part_o [((h*T + t)*nparts + p)*64 + d]     un-normalised sum_k exp(s_k - m) v_k[d]
//   part_ml[((h*T + t)*nparts + p)*2 + {0,1}]  m = max score of the chunk (or -1e30), l = sum_k exp(s_k - m)
// ---------------------------------------------------------------------------------------------------
struct FDArgs {
    dtensor q, k, v, m;
    int has_mask; float scale;
    int T, n_kv, H, rk2, rv2, nparts;
    float * part_o; float * part_ml;
};

template <int T>
__global__ void __launch_bounds__(256) k_fattn_dec(const FDArgs a) {
    __shared__ __attribute__((aligned(16))) float wo[4][T][64];
    __shared__ float wml[4][T][2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kg = lane >> 3, dc = lane & 7;                       // key group 0..7, dim chunk (8 dims = 16 bytes)
    const int hq = blockIdx.y, hk = hq / a.rk2, hv = hq / a.rv2;
    const int p = blockIdx.x;
    const int kbeg = p*128 + wave*32;

    // all K and V rows of this wave are requested up front: 8 x 16 B per lane in flight
    const char * kbase = a.k.data + (int64_t) hk*a.k.nb[2] + dc*16;
    const char * vbase = a.v.data + (int64_t) hv*a.v.nb[2] + dc*16;
    // (clamped key index instead of a predicate: a predicated load is waited for at the end of its basic block)
    uint4 kr[4], vr[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int key = kbeg + kg + 8*i, kc = key < a.n_kv ? key : a.n_kv - 1;
        kr[i] = *(const uint4 *) (kbase + (int64_t) kc*a.k.nb[1]);
        vr[i] = *(const uint4 *) (vbase + (int64_t) kc*a.v.nb[1]);
    }
    // q (rounded to f16 like the CPU's q_to_vec_dot), this lane's 8 dims of every query
    float qf[T][8];
    #pragma unroll
    for (int t = 0; t < T; t++) {
        const float * qp = (const float *) (a.q.data + (int64_t) t*a.q.nb[1] + (int64_t) hq*a.q.nb[2]) + dc*8;
        const float4 q0 = *(const float4 *) qp, q1 = *(const float4 *) (qp + 4);
        qf[t][0] = round_f16(q0.x); qf[t][1] = round_f16(q0.y); qf[t][2] = round_f16(q0.z); qf[t][3] = round_f16(q0.w);
        qf[t][4] = round_f16(q1.x); qf[t][5] = round_f16(q1.y); qf[t][6] = round_f16(q1.z); qf[t][7] = round_f16(q1.w);
    }
    // mask values of this lane's keys: requested together with everything else (a dependent load later would be a
    // serialized miss: the mask was written by the preceding cast kernel, on another XCD)
    uint16_t mkh[T][4];
    const char * mbase = a.has_mask ? a.m.data : (const char *) a.k.data;       // dummy (valid) address without a mask
    const int64_t mnb1 = a.has_mask ? a.m.nb[1] : 0;
    #pragma unroll
    for (int t = 0; t < T; t++)
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const int key = kbeg + kg + 8*i, kc = key < a.n_kv ? key : a.n_kv - 1;
            mkh[t][i] = *(const uint16_t *) (mbase + (int64_t) t*mnb1 + (int64_t) kc*2);
        }
    float sc[T][4];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int key = kbeg + kg + 8*i;
        const uint32_t w[4] = { kr[i].x, kr[i].y, kr[i].z, kr[i].w };
        float kf[8];
        #pragma unroll
        for (int e = 0; e < 4; e++) { kf[2*e] = h2f((uint16_t) (w[e] & 0xFFFF)); kf[2*e+1] = h2f((uint16_t) (w[e] >> 16)); }
        #pragma unroll
        for (int t = 0; t < T; t++) {
            float s = 0.0f;
            #pragma unroll
            for (int e = 0; e < 8; e++) s = fmaf(kf[e], qf[t][e], s);
            s = group_sum<8>(s);
            const float x = s * a.scale + (a.has_mask ? h2f(mkh[t][i]) : 0.0f);
            sc[t][i] = key < a.n_kv ? x : -INFINITY;
        }
    }
    #pragma unroll
    for (int t = 0; t < T; t++) {
        float m = fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3]));
        m = stride8_max(m);
        m = fmaxf(m, -1e30f);
        float l = 0.0f, o[8];
        #pragma unroll
        for (int e = 0; e < 8; e++) o[e] = 0.0f;
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const float pk = __expf(sc[t][i] - m);
            l += pk;
            const uint32_t w[4] = { vr[i].x, vr[i].y, vr[i].z, vr[i].w };
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                o[2*e]   = fmaf(pk, h2f((uint16_t) (w[e] & 0xFFFF)), o[2*e]);
                o[2*e+1] = fmaf(pk, h2f((uint16_t) (w[e] >> 16)),    o[2*e+1]);
            }
        }
        // sum over the 8 key groups (lanes with equal dc)
        l = stride8_sum(l);
        #pragma unroll
        for (int e = 0; e < 8; e++) o[e] = stride8_sum(o[e]);
        if (kg == 0) {
            *(float4 *) &wo[wave][t][dc*8]     = make_float4(o[0], o[1], o[2], o[3]);
            *(float4 *) &wo[wave][t][dc*8 + 4] = make_float4(o[4], o[5], o[6], o[7]);
            if (dc == 0) { wml[wave][t][0] = m; wml[wave][t][1] = l; }
        }
    }
    __syncthreads();
    // merge the 4 waves: thread (t, d) for t*64 + d < T*64
    for (int idx = tid; idx < T*64; idx += 256) {
        const int t = idx >> 6, d = idx & 63;
        const float m0 = wml[0][t][0], m1 = wml[1][t][0], m2 = wml[2][t][0], m3 = wml[3][t][0];
        const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float w0 = __expf(m0 - M), w1 = __expf(m1 - M), w2 = __expf(m2 - M), w3 = __expf(m3 - M);
        const float o = fmaf(w3, wo[3][t][d], fmaf(w2, wo[2][t][d], fmaf(w1, wo[1][t][d], w0 * wo[0][t][d])));
        const int64_t rec = ((int64_t) hq*T + t) * a.nparts + p;
        a.part_o[rec*64 + d] = o;
        if (d == 0) {
            a.part_ml[rec*2]     = M;
            a.part_ml[rec*2 + 1] = fmaf(w3, wml[3][t][1], fmaf(w2, wml[2][t][1], fmaf(w1, wml[1][t][1], w0 * wml[0][t][1])));
        }
    }
}


struct FC2Args { const float * part_o; const float * part_ml; int nparts, T, H; dtensor d; };
__global__ void __launch_bounds__(64) k_fattn_combine2(const FC2Args a) {
    const int t = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int64_t base = ((int64_t) h*a.T + t) * a.nparts;
    float M = -1e30f;
    for (int p = 0; p < a.nparts; p++) M = fmaxf(M, a.part_ml[(base + p)*2]);
    float L = 0.0f, O = 0.0f;
    for (int p = 0; p < a.nparts; p++) {
        const float w = __expf(a.part_ml[(base + p)*2] - M);
        L = fmaf(w, a.part_ml[(base + p)*2 + 1], L);
        O = fmaf(w, a.part_o[(base + p)*64 + lane], O);
    }
    float * dp = (float *) (a.d.data + (int64_t) h*a.d.nb[1] + (int64_t) t*a.d.nb[2]);
    dp[lane] = L == 0.0f ? 0.0f : O * (1.0f / L);
}

static int fattn_dec_check(const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v, const mi355x_tensor * mask) {
    if (q->type != MI355X_TYPE_F32 || k->type != MI355X_TYPE_F16 || v->type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
    if (q->ne[0] != 64 || k->ne[0] != 64 || v->ne[0] != 64 || q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return MI355X_E_UNSUPPORTED;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2) return MI355X_E_UNSUPPORTED;
    const int64_t T = q->ne[1], H = q->ne[2], n_kv = k->ne[1];
    if (T < 1 || T > 8 || H < 1 || n_kv < 1 || v->ne[1] != n_kv) return MI355X_E_UNSUPPORTED;
    if (k->ne[2] <= 0 || H % k->ne[2] || v->ne[2] <= 0 || H % v->ne[2]) return MI355X_E_UNSUPPORTED;
    if (((uintptr_t) q->data | q->nb[1] | q->nb[2]) % 16 || ((uintptr_t) k->data | k->nb[1] | k->nb[2]) % 16 ||
        ((uintptr_t) v->data | v->nb[1] | v->nb[2]) % 16) return MI355X_E_UNSUPPORTED;
    if (mask && (mask->type != MI355X_TYPE_F16 || mask->ne[0] < n_kv || mask->ne[1] < T || mask->nb[0] != 2 || mask->ne[2] != 1 || mask->ne[3] != 1)) return MI355X_E_UNSUPPORTED;
    return 0;
}

extern "C" int mi355x_flash_attn_partial(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                                         const mi355x_tensor * mask, float scale, mi355x_attn_partials * out) {
    const int rc0 = fattn_dec_check(q, k, v, mask);
    if (rc0) return rc0;
    const int T = (int) q->ne[1], H = (int) q->ne[2], n_kv = (int) k->ne[1];
    FDArgs a; memset(&a, 0, sizeof(a));
    a.q = to_d(q); a.k = to_d(k); a.v = to_d(v);
    if (mask) a.m = to_d(mask);
    a.has_mask = mask != nullptr; a.scale = scale; a.T = T; a.n_kv = n_kv; a.H = H;
    a.rk2 = (int) (H / k->ne[2]); a.rv2 = (int) (H / v->ne[2]);
    a.nparts = (n_kv + 127) / 128;
    mi355x_scratch_reset(ctx);
    const size_t nrec = (size_t) H * T * a.nparts;
    a.part_o  = (float *) mi355x_scratch_alloc(ctx, nrec * 64 * 4);
    a.part_ml = (float *) mi355x_scratch_alloc(ctx, nrec * 2 * 4);
    if (!a.part_o || !a.part_ml) return (int) hipErrorOutOfMemory;
    const dim3 grid(a.nparts, H), block(256);
    const double bytes = 2.0 * n_kv * 64 * 2 * H + (double) T*H*64*4 + (double) nrec*66*4;
    const double flops = 4.0 * T * (double) n_kv * 64 * H;
    int rc;
    switch (T) {
        case 1: rc = emit(ctx, "fattn_dec", k_fattn_dec<1>, grid, block, 0, a, bytes, flops); break;
        case 2: rc = emit(ctx, "fattn_dec", k_fattn_dec<2>, grid, block, 0, a, bytes, flops); break;
        case 3: rc = emit(ctx, "fattn_dec", k_fattn_dec<3>, grid, block, 0, a, bytes, flops); break;
        case 4: rc = emit(ctx, "fattn_dec", k_fattn_dec<4>, grid, block, 0, a, bytes, flops); break;
        case 5: rc = emit(ctx, "fattn_dec", k_fattn_dec<5>, grid, block, 0, a, bytes, flops); break;
        case 6: rc = emit(ctx, "fattn_dec", k_fattn_dec<6>, grid, block, 0, a, bytes, flops); break;
        case 7: rc = emit(ctx, "fattn_dec", k_fattn_dec<7>, grid, block, 0, a, bytes, flops); break;
        default: rc = emit(ctx, "fattn_dec", k_fattn_dec<8>, grid, block, 0, a, bytes, flops); break;
    }
    if (rc) return rc;
    out->part_o = a.part_o; out->part_ml = a.part_ml; out->nparts = a.nparts; out->T = T; out->H = H;
    return 0;
}

extern "C" int mi355x_flash_attn_combine(mi355x_ctx * ctx, const mi355x_attn_partials * p, const mi355x_tensor * dst) {
    if (dst->type != MI355X_TYPE_F32 || dst->ne[0] != 64 || dst->ne[1] != p->H || dst->ne[2] != p->T || dst->nb[0] != 4) return MI355X_E_UNSUPPORTED;
    FC2Args c = { p->part_o, p->part_ml, p->nparts, p->T, p->H, to_d(dst) };
    return emit(ctx, "fattn_combine", k_fattn_combine2, dim3(p->T, p->H), dim3(64), 0, c, 0, 0);
}
