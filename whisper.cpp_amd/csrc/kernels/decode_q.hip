// Decoder steps with 3..8 columns, and cross-state batches: the pre-quantized-activation pipeline.
//
// The fused T <= 2 kernels of decode.hip repeat the activation prologue (LayerNorm or attention combine, then Q8_0 / Q8_K
// quantization) in every one of their 256-1024 workgroups; with T columns that prologue — and the T*K*4 bytes every workgroup
// pulls through L2 for it — grows with T while the weight bytes stay the same (r02: 12.7 us for a 1280x1280 projection at T = 5
// against 4.0 us at T = 1, fc2 19.7 us).  Here a stage's activations are quantized ONCE:
//   k_act_prepare : one workgroup per column: [LayerNorm | combine of attention partial records | plain] -> Q8_0 / Q8_K planes
//                   in HBM, byte for byte the image the fused kernels build in LDS (decode_common.h)
//   k_gemv_q      : mat-vec over such planes: the prologue is a straight copy of the image into LDS (T*K*1.25 bytes), then
//                   the same per-lane block dots, reductions and epilogue as k_gemv_row.  Its POUT form (32 rows per
//                   workgroup) also leaves the Q8_0 planes of ITS result for the next mat-vec (fc1 + GELU -> fc2), so that
//                   stage needs no k_act_prepare at all
//   k_fattn_dec_multi / k_decode_head_multi : attention and step head for S states in one launch (cross-state batches)
// Per column the arithmetic is the fused kernels' own — same quantizer functions, same lane -> block assignment, same reduction
// trees — so a column's value does not depend on T, on its position, or on whether it ran here or in decode.hip
// (tests/test_gpu.py::test_plane_pipeline_is_bit_identical_to_the_fused_kernels).
//
// Reference arithmetic: activations -> Q8_0 / Q8_K (arch/x86/quants.c:302-398, ggml-quants.c:2768-2805), integer block dots with
// f32 scale-accumulate (ggml-cpu/quants.c:225-259, :365-406, :451-479, :696-769), ggml_norm (ggml-cpu/ops.cpp:3698-3765),
// flash_attn_ext final normalisation (ggml-cpu/ops.cpp:8479-8715), get_rows (ops.cpp:4850-5017).
#include "decode_common.h"

// ---------------------------------------------------------------------------------------------------
// k_act_prepare
// ---------------------------------------------------------------------------------------------------
struct APArgs {
    const float * xcol[MI355X_MAX_COLS];
    int K, T; float eps; int nparts;
    const float * ln_w; const float * ln_b;
    const float * part_o; const float * part_ml;
    void * planes;
};

// plane pointers of the (K, T) image
template <bool Q4K>
__device__ __forceinline__ void planes_of(void * base, int K, int T, uint32_t * & lo, uint32_t * & hi, float * & dx, int * & sx) {
    if constexpr (Q4K) {
        lo = (uint32_t *) base; hi = nullptr;
        dx = (float *) ((char *) base + (size_t) T*K);
        sx = (int *) (dx + T*(K >> 8));
    } else {
        const int nb = K >> 5;
        lo = (uint32_t *) base; hi = lo + (size_t) T*nb*4;
        dx = (float *) (hi + (size_t) T*nb*4);
        sx = (int *) (dx + T*nb);
    }
}

// MODE 0 plain, 1 LayerNorm + affine, 2 combine of attention partial records.  XS = float4 slots per thread (1: K <= 2048 with
// the block shape of the fused kernels, 64 * gemv_row_waves(K) threads; 5: K <= 5120 with 256 threads, MODE 0 only).
// One workgroup per column.
template <bool Q4K, int MODE, int XS>
__global__ void __launch_bounds__(512) k_act_prepare(const APArgs a) {
    __shared__ float red[2][8];
    // MODE 2 has no reduction wider than a wave: one WAVE per (column, 256 features) — blockIdx.y — instead of one workgroup per column, so that
    // 16 .. 32 columns' records (1 .. 2 MB) are combined by 80 .. 160 workgroups instead of 16 .. 32 (6.7 -> ~4.5 us at 16 columns)
    const int tid = threadIdx.x + (MODE == 2 ? blockIdx.y * blockDim.x : 0), wave = threadIdx.x >> 6, lane = tid & 63;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int K = a.K, K4 = K >> 2, T = a.T;
    const int t = blockIdx.x;
    // column t lives in image t / 8 (of min(8, T - 8 (t / 8)) columns) as column t % 8
    const int ti = t & (MI355X_IMG_COLS - 1), Ti = T - (t - ti) < MI355X_IMG_COLS ? T - (t - ti) : MI355X_IMG_COLS;
    uint32_t * lo, * hi; float * dx; int * sx;
    planes_of<Q4K>((char *) a.planes + (size_t) (t >> 3) * dg_img_stride(Q4K ? MI355X_TYPE_Q4_K : MI355X_TYPE_Q8_0, K), K, Ti, lo, hi, dx, sx);
    auto act_store = [&](const float v[4], int e) {
        if constexpr (Q4K) dg_q8_K_store(v, e, ti, K, Ti, lo, dx, sx);
        else               dg_q8_0_store(v, e, ti, K >> 5, lo, hi, dx, sx);
    };
    float4 xr[XS];
    if constexpr (MODE == 2) {
        // x[h*64 + d] = sum_p w_p o_p[d] / sum_p w_p l_p,  w_p = exp(m_p - max_p m_p): the statement sequence of k_gemv_row MODE 2
        const int pe4 = tid < K4 ? tid : K4 - 1, ph = pe4 >> 4, pd = (pe4 & 15) << 2;
        const int64_t base = ((int64_t) ph*T + t) * a.nparts;
        float M = -1e30f, L = 0.0f;
        float4 o = make_float4(0, 0, 0, 0);
        for (int p = 0; p < a.nparts; p++) M = fmaxf(M, a.part_ml[(base + p)*2]);
        for (int p = 0; p < a.nparts; p++) {
            const float2 ml = *(const float2 *) (a.part_ml + (base + p)*2);
            const float4 ov = *(const float4 *) (a.part_o + (base + p)*64 + pd);
            const float w = __expf(ml.x - M);
            L = fmaf(w, ml.y, L);
            o.x = fmaf(w, ov.x, o.x); o.y = fmaf(w, ov.y, o.y); o.z = fmaf(w, ov.z, o.z); o.w = fmaf(w, ov.w, o.w);
        }
        const float inv = L == 0.0f ? 0.0f : 1.0f / L;
        xr[0] = make_float4(o.x*inv, o.y*inv, o.z*inv, o.w*inv);
    } else {
        const char * xp = (const char *) a.xcol[t];
        #pragma unroll
        for (int i = 0; i < XS; i++) {
            const int e4 = tid + i*nthreads;
            xr[i] = *(const float4 *) (xp + (size_t) (e4 < K4 ? e4 : K4 - 1)*16);
        }
    }
    if constexpr (MODE == 1) {
        // ggml_norm + affine, the statement sequence of k_gemv_row MODE 1 for one column
        const int e4c = tid < K4 ? tid : K4 - 1;
        const float4 lw = *(const float4 *) (a.ln_w + e4c*4), lb = *(const float4 *) (a.ln_b + e4c*4);
        float p = 0.0f;
        if (tid < K4) p += (xr[0].x + xr[0].y) + (xr[0].z + xr[0].w);
        p = wave_sum(p);
        if (lane == 0) red[0][wave] = p;
        __syncthreads();
        float pw[8];
        #pragma unroll
        for (int w = 0; w < 8; w++) pw[w] = red[0][w];
        float rs = 0.0f;
        #pragma unroll
        for (int w = 0; w < 8; w++) rs += w < nwaves ? pw[w] : 0.0f;
        const float mean = rs / K;
        p = 0.0f;
        if (tid < K4) {
            const float d0 = xr[0].x - mean, d1 = xr[0].y - mean, d2 = xr[0].z - mean, d3 = xr[0].w - mean;
            p += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
        }
        p = wave_sum(p);
        if (lane == 0) red[1][wave] = p;
        __syncthreads();
        #pragma unroll
        for (int w = 0; w < 8; w++) pw[w] = red[1][w];
        rs = 0.0f;
        #pragma unroll
        for (int w = 0; w < 8; w++) rs += w < nwaves ? pw[w] : 0.0f;
        const float rstd = 1.0f / sqrtf(rs / K + a.eps);
        if (tid < K4) {
            float o[4] = { (xr[0].x - mean) * rstd, (xr[0].y - mean) * rstd, (xr[0].z - mean) * rstd, (xr[0].w - mean) * rstd };
            o[0] = o[0]*lw.x; o[1] = o[1]*lw.y; o[2] = o[2]*lw.z; o[3] = o[3]*lw.w;
            o[0] = o[0]+lb.x; o[1] = o[1]+lb.y; o[2] = o[2]+lb.z; o[3] = o[3]+lb.w;
            act_store(o, tid*4);
        }
    } else {
        #pragma unroll
        for (int i = 0; i < XS; i++) {
            const int e4 = tid + i*nthreads;
            if (e4 < K4) {
                const float v[4] = { xr[i].x, xr[i].y, xr[i].z, xr[i].w };
                act_store(v, e4*4);
            }
        }
    }
}

extern "C" size_t mi355x_act_planes_bytes(int wtype, int K, int T) { return dg_planes_bytes(wtype, K, T); }

extern "C" void * mi355x_act_scratch(mi355x_ctx * ctx, int which) {
    const size_t each = (size_t) 384 << 10;          // >= planes of K = 8192, T = 32 (4 images of 83 KB), read in whole 16-byte words
    if (!ctx->qact) {
        (void) hipSetDevice(ctx->device);
        if (hipMalloc(&ctx->qact, 2 * each) != hipSuccess) { (void) hipGetLastError(); ctx->qact = nullptr; return nullptr; }
        // zero-fill ON THE KERNELS' STREAM: a hipMemset on the null stream is not ordered with this (non-blocking) stream and may land while
        // the first chain already uses the planes (round 3: the first multi-token step of a process came out wrong at random)
        (void) hipMemsetAsync(ctx->qact, 0, 2 * each, ctx->stream);
    }
    return (char *) ctx->qact + (which & 1) * each;
}

extern "C" int mi355x_act_prepare(mi355x_ctx * ctx, const mi355x_act_desc * d, void * planes) {
    const int K = d->K, T = d->T, wt = d->wtype;
    if (T < 1 || T > MI355X_MAX_COLS || !planes || ((uintptr_t) planes % 16)) return MI355X_E_UNSUPPORTED;
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0 && wt != MI355X_TYPE_Q4_K) return MI355X_E_UNSUPPORTED;
    if (K <= 0 || K % 32 || (wt == MI355X_TYPE_Q4_K && K % 256) || K > 8192) return MI355X_E_UNSUPPORTED;
    const bool q4k = wt == MI355X_TYPE_Q4_K;
    APArgs a; memset(&a, 0, sizeof(a));
    a.K = K; a.T = T; a.eps = d->eps; a.planes = planes;
    const bool from_part = !d->x && !d->xcol[0];
    int mode = 0;
    if (from_part) {
        if (!d->attn_part_o || !d->attn_part_ml || d->attn_nparts < 1 || K % 64 || K > 2048 || d->has_norm) return MI355X_E_UNSUPPORTED;
        a.part_o = d->attn_part_o; a.part_ml = d->attn_part_ml; a.nparts = d->attn_nparts;
        mode = 2;
    } else {
        for (int t = 0; t < T; t++) {
            a.xcol[t] = d->xcol[0] ? d->xcol[t] : (const float *) ((const char *) d->x + (int64_t) t*d->x_nb1);
            if (!a.xcol[t] || ((uintptr_t) a.xcol[t] % 16)) return MI355X_E_UNSUPPORTED;
        }
        if (d->has_norm) {
            if (K > 2048 || !d->ln_w || !d->ln_b || ((uintptr_t) d->ln_w % 16) || ((uintptr_t) d->ln_b % 16)) return MI355X_E_UNSUPPORTED;
            a.ln_w = d->ln_w; a.ln_b = d->ln_b;
            mode = 1;
        }
    }
    const double bytes = (double) T * K * 4 + (double) dg_planes_bytes(wt, K, T);
    const dim3 grid(T, mode == 2 ? (K / 4 + 63) / 64 : 1);
    if (K <= 2048) {
        const dim3 block(mode == 2 ? 64 : 64 * gemv_row_waves(K));
        if (q4k) switch (mode) {
            case 0:  return emit(ctx, "act_prepare", k_act_prepare<true, 0, 1>, grid, block, 0, a, bytes, 0);
            case 1:  return emit(ctx, "act_prepare", k_act_prepare<true, 1, 1>, grid, block, 0, a, bytes, 0);
            default: return emit(ctx, "act_prepare", k_act_prepare<true, 2, 1>, grid, block, 0, a, bytes, 0);
        }
        switch (mode) {
            case 0:  return emit(ctx, "act_prepare", k_act_prepare<false, 0, 1>, grid, block, 0, a, bytes, 0);
            case 1:  return emit(ctx, "act_prepare", k_act_prepare<false, 1, 1>, grid, block, 0, a, bytes, 0);
            default: return emit(ctx, "act_prepare", k_act_prepare<false, 2, 1>, grid, block, 0, a, bytes, 0);
        }
    }
    if (mode != 0) return MI355X_E_UNSUPPORTED;
    // K <= 8192: 256 threads x 8 slots (a Q8_K super-block is one wave's 64 float4 of one slot: 256 threads keep that alignment)
    if (q4k) return emit(ctx, "act_prepare", k_act_prepare<true, 0, 8>, grid, dim3(256), 0, a, bytes, 0);
    return emit(ctx, "act_prepare", k_act_prepare<false, 0, 8>, grid, dim3(256), 0, a, bytes, 0);
}

// ---------------------------------------------------------------------------------------------------
// k_gemv_q
// ---------------------------------------------------------------------------------------------------
struct QSeg {
    const void * w; int64_t nbt; int N; int has_scale;
    const float * bias; float scale; int gelu;
    int dst_f16; int pad;
};
struct QGArgs {
    const void * planes; int K, T, nseg, ntot;
    int row_start[4];
    QSeg seg[3];
    const uint16_t * gelu_tab;
    void * planes_out; int planes_only; int pad;
    mi355x_gemv_cols cols;
};

// dot of one 64-element Q4_K weight unit with the Q8_K activations of columns 0..TMAX-1 (columns >= T read column T-1 again and are
// never stored): wblk_dot_q4k of decode_common.h with a run-time column count in the plane strides
template <int TMAX>
__device__ __forceinline__ void wblk_dot_q4k_rt(const wblk<MI355X_TYPE_Q4_K> & r, int ch, float live, int nch, int nsb, int T,
                                                const uint4 * pl, const float * dx, const int * bs, float * acc, float * accm) {
    const int sb = ch >> 2, c = ch & 3;
    const float dw = h2f((uint16_t) (r.dm & 0xFFFF)) * live, dminw = h2f((uint16_t) (r.dm >> 16)) * live;
    int sc_lo, m_lo, sc_hi, m_hi;
    q4k_scale_min_w(2*c,     r.sc[0], r.sc[1], r.sc[2], sc_lo, m_lo);
    q4k_scale_min_w(2*c + 1, r.sc[0], r.sc[1], r.sc[2], sc_hi, m_hi);
    const uint32_t w[8] = { r.q[0], r.q[1], r.q[2], r.q[3], r.q1[0], r.q1[1], r.q1[2], r.q1[3] };
    #pragma unroll
    for (int tt = 0; tt < TMAX; tt++) {
        const int t = tt < T ? tt : T - 1;
        const uint4 a0 = pl[((size_t) 0*T + t)*nch + ch], a1 = pl[((size_t) 1*T + t)*nch + ch];
        const uint4 a2 = pl[((size_t) 2*T + t)*nch + ch], a3 = pl[((size_t) 3*T + t)*nch + ch];
        const uint32_t al[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
        const uint32_t ah[8] = { a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w };
        int dlo = 0, dhi = 0;
        #pragma unroll
        for (int i = 0; i < 8; i++) {
            dlo = __builtin_amdgcn_sdot4((int) (w[i] & 0x0F0F0F0Fu),        (int) al[i], dlo, false);
            dhi = __builtin_amdgcn_sdot4((int) ((w[i] >> 4) & 0x0F0F0F0Fu), (int) ah[i], dhi, false);
        }
        const int isum = sc_lo*dlo + sc_hi*dhi;
        const int msum = m_lo*bs[t*(nsb*8) + sb*8 + 2*c] + m_hi*bs[t*(nsb*8) + sb*8 + 2*c + 1];
        const float dxv = dx[t*nsb + sb];
        acc[tt]  = fmaf(dxv*dw, (float) isum, acc[tt]);
        accm[tt] = fmaf(-dxv*dminw, (float) msum, accm[tt]);
    }
}

// TMAX: columns the kernel is built for (run-time T <= TMAX).  NU: lane-units per row and lane (1: K <= 2048, 3: K <= 6144; Q4_K
// units are 64 elements: 1: K <= 4096, 2: K <= 8192).  R rows per wave.  POUT: 8 waves x 4 rows = the 32 rows of one Q8_0
// block of the RESULT per workgroup, whose planes are written as well (single segment).
// MG: more than one image (T > 8).  MG = false is the single-image kernel: no image loop, no prefetch registers.
template <int WT, int TMAX, int NU, bool NSEG1, int R, bool POUT, bool MG>
__global__ void __launch_bounds__(512) k_gemv_q(const QGArgs a) {
    constexpr bool Q4K = WT == MI355X_TYPE_Q4_K;
    constexpr int CP = (TMAX * (Q4K ? NU*4096*9/8 + 256 : NU*64*40) / 16 + 255) / 256;       // uint4 copy slots per thread (>= 256 threads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int K = a.K, nb = Q4K ? K >> 6 : K >> 5, nsb = K >> 8;
    // T columns = G images of <= 8 columns (TMAX < 8: one image).  The weights are loaded once and stay in registers; the images pass
    // through LDS one after the other (the next one is requested while the current one is used).
    const int Tall = a.T, G = MG ? (Tall + MI355X_IMG_COLS - 1) / MI355X_IMG_COLS : 1;
    const size_t istride = dg_img_stride(WT, K);
    int T = Tall < MI355X_IMG_COLS ? Tall : MI355X_IMG_COLS;          // columns of the current image
    const int ntot = a.ntot;
    const int grow = __builtin_amdgcn_readfirstlane((blockIdx.x * nwaves + wave) * R);
    int s = 0;
    if constexpr (!NSEG1) {
        if (a.nseg > 1 && grow >= a.row_start[1]) s = 1;
        if (a.nseg > 2 && grow >= a.row_start[2]) s = 2;
    }
    const bool rok = grow < ntot;
    const int row = rok ? grow - (NSEG1 ? 0 : a.row_start[s]) : 0;
    const QSeg & sgr = NSEG1 ? a.seg[0] : a.seg[s];
    struct { const void * w; int64_t nbt; const float * bias; float scale; int has_scale, gelu, dst_f16; } sg;
    sg.w = sgr.w; sg.nbt = sgr.nbt; sg.bias = sgr.bias; sg.scale = sgr.scale; sg.has_scale = sgr.has_scale; sg.gelu = sgr.gelu; sg.dst_f16 = sgr.dst_f16;
    // this lane's column in the epilogue: lane r*TMAX + t finishes (row r, column t of the image)
    const int tl = lane % TMAX;
    const int rlane = lane / TMAX < R ? lane / TMAX : R - 1;
    // per-column pointer tables: read straight from the kernarg segment (constant address space, scalar loads).  Taking the
    // address of the by-value argument `a` instead would force a private (scratch) copy of it.
    typedef const QGArgs __attribute__((address_space(4))) * kargs_t;
    const kargs_t ka = (kargs_t) __builtin_amdgcn_kernarg_segment_ptr();
    const int sc = NSEG1 ? 0 : s;
    // ---- destination / residual column of the lane in image gi_ (T_ columns) ----
    int tcol; void * dcol; const float * rcol;
    auto pick_cols = [&](int gi_, int T_) {
        tcol = tl < T_ ? tl : T_ - 1;
        const int c0 = gi_ * MI355X_IMG_COLS;
        void * d0 = ka->cols.dst[sc][c0]; const float * r0 = ka->cols.res[sc][c0];
        dcol = d0; rcol = r0;
        #pragma unroll
        for (int t = 1; t < TMAX; t++) {
            void * dv = ka->cols.dst[sc][c0 + t]; const float * rv = ka->cols.res[sc][c0 + t];
            asm volatile("" :: "s"(dv), "s"(rv));
            dcol = tcol == t ? dv : dcol; rcol = tcol == t ? rv : rcol;
        }
        asm volatile("" :: "s"(d0), "s"(r0));
    };
    pick_cols(0, T);
    asm volatile("" :: "s"(sg.w), "s"(sg.nbt), "s"(sg.bias), "s"(sg.scale), "s"(sg.has_scale), "s"(sg.gelu), "s"(sg.dst_f16), "s"(a.gelu_tab), "s"(a.planes));

    // ---- the load burst: plane image (L2), bias / residual, weights (HBM) last; clamped addresses, no predicates ----
    auto img_n16 = [&](int Ti) { return (int) (((Q4K ? (size_t) Ti * ((size_t) K + nsb*4 + (K >> 5)*4) : (size_t) Ti * nb * 40) + 15) >> 4); };
    int n16 = img_n16(T);
    u32x4 cp[CP];
    #pragma unroll
    for (int i = 0; i < CP; i++) {
        const int idx = tid + i*nthreads;
        cp[i] = ((const u32x4 *) a.planes)[idx < n16 ? idx : n16 - 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    const float * bptr = sg.bias ? sg.bias + row + rlane : (const float *) a.gelu_tab;
    const float * rptr = rcol ? rcol + row + rlane : (const float *) a.gelu_tab;
    const float bias_v = *bptr;
    float res_v = *rptr;
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

    for (int gi = 0; ; gi++) {
        // ---- plane image -> LDS ----
        // (unconditional stores: surplus slots land in one dummy word behind the image.  A predicated store in this loop makes the
        //  compiler turn the predicate into a loop exit, keep the loop rolled and park cp[] in scratch memory.)
        #pragma unroll
        for (int i = 0; i < CP; i++) {
            const int idx = tid + i*nthreads;
            ((u32x4 *) smem)[idx < n16 ? idx : n16] = cp[i];
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);          // nothing that needs the WEIGHTS may move above the barrier: they are still in flight while the planes settle
        const int Tn = Tall - (gi + 1) * MI355X_IMG_COLS < MI355X_IMG_COLS ? Tall - (gi + 1) * MI355X_IMG_COLS : MI355X_IMG_COLS;      // columns of the next image
        if constexpr (MG) {
            if (gi + 1 < G) {           // the next image: in flight under this image's dots
                n16 = img_n16(Tn);
                const u32x4 * nxt = (const u32x4 *) ((const char *) a.planes + (size_t) (gi + 1) * istride);
                #pragma unroll
                for (int i = 0; i < CP; i++) {
                    const int idx = tid + i*nthreads;
                    cp[i] = nxt[idx < n16 ? idx : n16 - 1];
                }
            }
        }
        const uint4 * alo = (const uint4 *) smem;
        const uint4 * ahi = alo + (size_t) T*nb;
        const float * dx = Q4K ? (const float *) (smem + (size_t) T*K) : (const float *) (ahi + (size_t) T*nb);
        const int *   sx = Q4K ? (const int *) (dx + T*nsb) : (const int *) (dx + T*nb);

        // ---- dot products: lane handles units lane (+64, +128) of the wave's rows; clamped duplicates carry weight 0 ----
        float acc[R][TMAX];
        #pragma unroll
        for (int r = 0; r < R; r++)
            #pragma unroll
            for (int t = 0; t < TMAX; t++) acc[r][t] = 0.0f;
        if constexpr (Q4K) {
            #pragma unroll
            for (int r = 0; r < R; r++) {
                float accm[TMAX];
                #pragma unroll
                for (int t = 0; t < TMAX; t++) accm[t] = 0.0f;
                #pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int g = lane + 64*u, gc = g < nb ? g : nb - 1;
                    wblk_dot_q4k_rt<TMAX>(wr[r][u], gc, g < nb ? 1.0f : 0.0f, nb, nsb, T, alo, dx, sx, acc[r], accm);
                }
                #pragma unroll
                for (int t = 0; t < TMAX; t++) acc[r][t] += accm[t];
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
                for (int tt = 0; tt < TMAX; tt++) {
                    const int t = tt < T ? tt : T - 1;
                    const uint4 al = alo[(size_t) t*nb + gc], ah = ahi[(size_t) t*nb + gc];
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
                        acc[r][tt] = fmaf(dw[r] * dxv, (float) sum, acc[r][tt]);
                    }
                }
            }
        }
        #pragma unroll
        for (int r = 0; r < R; r++)
            #pragma unroll
            for (int t = 0; t < TMAX; t++) acc[r][t] = wave_sum(acc[r][t]);
        float v = acc[0][0];
        #pragma unroll
        for (int r = 0; r < R; r++)
            #pragma unroll
            for (int t = 0; t < TMAX; t++) if (r + t > 0) v = (lane == r*TMAX + t) ? acc[r][t] : v;
        const bool mine = rok && lane < R*TMAX && tl < T;
        if (mine) {
            if (sg.bias)      v = v + bias_v;
            if (sg.has_scale) v = v * sg.scale;
            if (sg.gelu)      v = gelu_lut(v, a.gelu_tab);
            if (rcol)         v = v + res_v;
        }
        if constexpr (POUT) {
            // the 32 rows of this workgroup are one Q8_0 block of the result for every column: tile[t][32] -> quantize -> planes of K' = ntot
            // (behind the largest image + its dummy word: the image of a later group never reaches it)
            float * tile = (float *) (smem + (((size_t) img_n16(Tall < MI355X_IMG_COLS ? Tall : MI355X_IMG_COLS) + 1) << 4));
            if (mine) tile[tcol*32 + wave*R + rlane] = v;
            if (mine && !a.planes_only) {
                if (sg.dst_f16) ((uint16_t *) dcol)[row + rlane] = f2h(v); else ((float *) dcol)[row + rlane] = v;
            }
            __syncthreads();
            if (tid < T*8) {
                const int t = tid >> 3, q = tid & 7;
                const float4 x4 = *(const float4 *) (tile + t*32 + q*4);
                const float xv[4] = { x4.x, x4.y, x4.z, x4.w };
                const int nbo = ntot >> 5;
                uint32_t * olo = (uint32_t *) ((char *) a.planes_out + (size_t) gi * dg_img_stride(MI355X_TYPE_Q8_0, ntot)), * ohi = olo + (size_t) T*nbo*4;
                float * odx = (float *) (ohi + (size_t) T*nbo*4);
                int * osx = (int *) (odx + T*nbo);
                dg_q8_0_store(xv, (int) blockIdx.x*32 + q*4, t, nbo, olo, ohi, odx, osx);
            }
        } else {
            if (mine) {
                if (sg.dst_f16) ((uint16_t *) dcol)[row + rlane] = f2h(v); else ((float *) dcol)[row + rlane] = v;
            }
        }
        if (!MG || gi + 1 >= G) break;
        __syncthreads();                        // every wave is done with this image (and with the result tile) before the next one lands
        T = Tn;
        pick_cols(gi + 1, T);
        res_v = *(rcol ? rcol + row + rlane : (const float *) a.gelu_tab);
    }
}

template <int WT, int TMAX, int NU, bool MG>
static int launch_gemv_q_v(mi355x_ctx * ctx, const QGArgs & k, bool nseg1, bool pout, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    const char * name = "gemv_q";
    if constexpr (NU == 1 && WT != MI355X_TYPE_Q4_K) {
        if (pout) return emit(ctx, name, k_gemv_q<WT, TMAX, NU, true, 4, true, MG>, grid, block, lds, k, bytes, flops);                        //  8 waves x 4 rows
    }
    if (pout) return MI355X_E_UNSUPPORTED;
    if (nseg1) return emit(ctx, name, k_gemv_q<WT, TMAX, NU, true, 1, false, MG>, grid, block, lds, k, bytes, flops);
    if constexpr (NU == 1) return emit(ctx, name, k_gemv_q<WT, TMAX, NU, false, 1, false, MG>, grid, block, lds, k, bytes, flops);
    return MI355X_E_UNSUPPORTED;
}
template <int WT>
static int launch_gemv_q(mi355x_ctx * ctx, const QGArgs & k, int nu, bool nseg1, bool pout, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    constexpr int NUBIG = WT == MI355X_TYPE_Q4_K ? 2 : 3;
    if (k.T <= 4) {
        if (nu == 1) return launch_gemv_q_v<WT, 4, 1, false>(ctx, k, nseg1, pout, grid, block, lds, bytes, flops);
        if (pout) return MI355X_E_UNSUPPORTED;
        return launch_gemv_q_v<WT, 4, NUBIG, false>(ctx, k, nseg1, false, grid, block, lds, bytes, flops);
    }
    if (k.T <= MI355X_IMG_COLS) {
        if (nu == 1) return launch_gemv_q_v<WT, 8, 1, false>(ctx, k, nseg1, pout, grid, block, lds, bytes, flops);
        if (pout) return MI355X_E_UNSUPPORTED;
        return launch_gemv_q_v<WT, 8, NUBIG, false>(ctx, k, nseg1, false, grid, block, lds, bytes, flops);
    }
    if (nu == 1) return launch_gemv_q_v<WT, 8, 1, true>(ctx, k, nseg1, pout, grid, block, lds, bytes, flops);
    if (pout) return MI355X_E_UNSUPPORTED;
    return launch_gemv_q_v<WT, 8, NUBIG, true>(ctx, k, nseg1, false, grid, block, lds, bytes, flops);
}

// mat-vec over pre-quantized activation planes; MI355X_E_UNSUPPORTED: the caller tries k_gemv8 (which copies the same image)
int mi355x_gemv_q(mi355x_ctx * ctx, const mi355x_gemv_desc * d) {
    if (!d->x_planes || d->x || d->attn_part_o || d->has_norm) return MI355X_E_UNSUPPORTED;
    if (d->nseg < 1 || d->nseg > 3 || d->T < 1 || d->T > MI355X_MAX_COLS || ((uintptr_t) d->x_planes % 16)) return MI355X_E_UNSUPPORTED;
    const int wt = d->seg[0].wtype, K = d->K, T = d->T;
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0 && wt != MI355X_TYPE_Q4_K) return MI355X_E_UNSUPPORTED;
    const bool q4k = wt == MI355X_TYPE_Q4_K;
    if (K <= 0 || K % 32 || (q4k && K % 256)) return MI355X_E_UNSUPPORTED;
    const int units = q4k ? K / 64 : K / 32;
    const int nu = units <= 64 ? 1 : (q4k ? 2 : 3);
    if (units > 64 * nu) return MI355X_E_UNSUPPORTED;
    QGArgs k; memset(&k, 0, sizeof(k));
    k.planes = d->x_planes; k.K = K; k.T = T; k.nseg = d->nseg; k.gelu_tab = ctx->gelu_tab;
    int ntot = 0; double wbytes = 0;
    for (int s = 0; s < d->nseg; s++) {
        const mi355x_gemv_seg & g = d->seg[s];
        if (g.wtype != wt || g.N <= 0 || ((uintptr_t) g.w % 16)) return MI355X_E_UNSUPPORTED;
        if (g.dst_type != MI355X_TYPE_F32 && g.dst_type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
        k.row_start[s] = ntot;
        QSeg & o = k.seg[s];
        o.w = g.w; o.N = g.N; o.nbt = q4k ? (int64_t) g.N * (K / 256) : (int64_t) g.N * (K / 32);
        o.bias = g.ep.bias; o.scale = g.ep.scale; o.has_scale = g.ep.has_scale; o.gelu = g.ep.gelu; o.dst_f16 = g.dst_type == MI355X_TYPE_F16;
        for (int t = 0; t < T; t++) {
            if (d->cols) { k.cols.dst[s][t] = d->cols->dst[s][t]; k.cols.res[s][t] = d->cols->res[s][t]; }
            else {
                k.cols.dst[s][t] = g.dst ? (char *) g.dst + (int64_t) t*g.dst_nb1 : nullptr;
                k.cols.res[s][t] = g.ep.residual ? (const float *) ((const char *) g.ep.residual + (int64_t) t*g.ep.residual_nb1) : nullptr;
            }
            if (!k.cols.dst[s][t] && !(d->planes_out && d->planes_out_only)) return MI355X_E_UNSUPPORTED;
            // a segment either has a residual in every column or in none
            if ((k.cols.res[s][t] != nullptr) != (k.cols.res[s][0] != nullptr)) return MI355X_E_UNSUPPORTED;
        }
        for (int t = T; t < MI355X_MAX_COLS; t++) { k.cols.dst[s][t] = k.cols.dst[s][T - 1]; k.cols.res[s][t] = k.cols.res[s][T - 1]; }
        ntot += g.N;
        wbytes += (double) mi355x_type_row_bytes(wt, K) * g.N;
    }
    for (int s = d->nseg; s < 4; s++) k.row_start[s] = ntot;
    if (ntot > 8192) return MI355X_E_UNSUPPORTED;                  // the vocabulary projection: k_gemv8
    k.ntot = ntot;
    const bool pout = d->planes_out != nullptr;
    if (pout) {
        if (q4k || d->nseg != 1 || ntot % 32 || nu != 1 || ((uintptr_t) d->planes_out % 16)) return MI355X_E_UNSUPPORTED;
        k.planes_out = d->planes_out; k.planes_only = d->planes_out_only;
    }
    const size_t img = (dg_act_bytes(wt, K, T < MI355X_IMG_COLS ? T : MI355X_IMG_COLS) + 15) & ~(size_t) 15;      // one image of <= 8 columns at a time
    const size_t lds = img + 16 + (pout ? (size_t) MI355X_IMG_COLS * 32 * 4 : 0);      // image | dummy word | result tile
    if (lds > 64 * 1024) return MI355X_E_UNSUPPORTED;
    // the 32 rows of a planes-out workgroup: 8 waves x 4 rows (16 waves x 2 rows measured no faster)
    const int prows = 4;
    const int waves = pout ? 32 / prows : gemv_row_waves(K), rpb = pout ? 32 : waves;
    const dim3 grid((ntot + rpb - 1) / rpb), block(64 * waves);
    const double bytes = wbytes + (double) dg_planes_bytes(wt, K, T) + (double) ntot*T*4;
    const double flops = 2.0 * ntot * K * T;
    switch (wt) {
        case MI355X_TYPE_Q4_0: return launch_gemv_q<MI355X_TYPE_Q4_0>(ctx, k, nu, d->nseg == 1, pout, grid, block, (uint32_t) lds, bytes, flops);
        case MI355X_TYPE_Q5_0: return launch_gemv_q<MI355X_TYPE_Q5_0>(ctx, k, nu, d->nseg == 1, pout, grid, block, (uint32_t) lds, bytes, flops);
        case MI355X_TYPE_Q8_0: return launch_gemv_q<MI355X_TYPE_Q8_0>(ctx, k, nu, d->nseg == 1, pout, grid, block, (uint32_t) lds, bytes, flops);
        case MI355X_TYPE_Q4_K: return launch_gemv_q<MI355X_TYPE_Q4_K>(ctx, k, nu, d->nseg == 1, pout, grid, block, (uint32_t) lds, bytes, flops);
    }
    return MI355X_E_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------
// k_vocab: the vocabulary projection (final LayerNorm + [n_vocab x K] mat-vec, src/whisper.cpp:2832-2844), N > 8192 rows
// ---------------------------------------------------------------------------------------------------
// The one decode kernel that is bandwidth-bound (45.6 MB for large-v3 Q5_0, read once per step).  k_gemv8 serves it with its generic
// machinery (segments, passes x chunks with divisions, epilogue options, predicated units: ~3300 instructions); here the same per-row
// arithmetic — 8 lanes per weight row, lane j takes units j, j + 8, ..., f32 fma chain in unit order, 3-step lane reduction — runs with
// nothing else around it:
//   * every wave owns row groups g, g + stride, ... (8 rows each) and has TWO of them in flight (both requested before the prologue,
//     a buffer is refilled as soon as it has been used); the launch is sized so that two groups per wave cover the matrix: all of it
//     is requested in the first microsecond;
//   * per lane one base pointer per plane, the units are immediate offsets, the next group is one scalar add;
//   * T = 1: the lane's activation blocks (the same for every row) are read from LDS once and kept in registers;
//   * the prologue is k_act_prepare's LayerNorm (same statements: the planes equal the ones a cross-state batch prepares) or a copy of
//     prepared planes.
// K % 256 == 0 (8 | K/32), K <= 2048, one segment, no epilogue options, F32 destinations (per column) + optional mirror.
struct VArgs {
    const void * w; int64_t nbt; int N, K, T, has_norm;
    const float * x; int64_t x_nb1; const float * ln_w; const float * ln_b; float eps; int pad;
    const void * xq;
    void * dstcol[MI355X_MAX_COLS]; void * mircol[MI355X_MAX_COLS];
};

// NG: images of 8 columns (T > 8: cross-state batches, prepared planes only); the row groups' weights are unpacked once and meet every image.
template <int WT, int TMAX, int U, int NG>
__global__ void __launch_bounds__(512) k_vocab(const VArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[2][TMAX][8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r8 = lane >> 3, j8 = lane & 7;
    const int K = a.K, Tall = a.T, N = a.N;
    const int T = NG == 1 ? Tall : MI355X_IMG_COLS;          // columns of image 0 (NG > 1: full)
    constexpr int nb = U * 8;
    const int ngroups = (N + 7) >> 3;
    const int stride = gridDim.x * 8;
    int grp = blockIdx.x * 8 + wave;

    // ---- loads: activations (L2) first, then two row groups of weights (HBM) ----
    const int K4 = K >> 2, e4c = tid < K4 ? tid : K4 - 1;
    float4 xr[TMAX], lw, lb;
    if (NG == 1 && a.has_norm) {
        #pragma unroll
        for (int tt = 0; tt < TMAX; tt++) xr[tt] = *(const float4 *) ((const char *) a.x + (int64_t) (tt < T ? tt : T - 1)*a.x_nb1 + (size_t) e4c*16);
        lw = *(const float4 *) (a.ln_w + e4c*4); lb = *(const float4 *) (a.ln_b + e4c*4);
    }
    wblk<WT> A[U], B[U];
    auto load_group = [&](wblk<WT> * buf, int g) {
        const int gc = g < ngroups ? g : ngroups - 1;
        int row = gc*8 + r8; row = row < N ? row : N - 1;
        const int64_t ib = (int64_t) row*nb + j8;
        #pragma unroll
        for (int u = 0; u < U; u++) wblk_load<WT>(buf[u], (const char *) a.w, a.nbt, ib + 8*u);
    };
    load_group(A, grp);
    load_group(B, grp + stride);
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: activation planes in LDS ----
    uint32_t * plo, * phi; float * pdx; int * psx;
    planes_of<false>(smem, K, T, plo, phi, pdx, psx);
    if (NG == 1 && a.has_norm) {
        // k_act_prepare MODE 1, all columns at once (blockDim 512: waves >= K4/64 contribute zeros)
        #pragma unroll
        for (int tt = 0; tt < TMAX; tt++) {
            float p = 0.0f;
            if (tid < K4) p += (xr[tt].x + xr[tt].y) + (xr[tt].z + xr[tt].w);
            p = wave_sum(p);
            if (lane == 0) red[0][tt][wave] = p;
        }
        __syncthreads();
        float mean[TMAX];
        #pragma unroll
        for (int tt = 0; tt < TMAX; tt++) {
            float rs = 0.0f;
            #pragma unroll
            for (int w = 0; w < 8; w++) rs += red[0][tt][w];
            mean[tt] = rs / K;
            float p = 0.0f;
            if (tid < K4) {
                const float d0 = xr[tt].x - mean[tt], d1 = xr[tt].y - mean[tt], d2 = xr[tt].z - mean[tt], d3 = xr[tt].w - mean[tt];
                p += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
            }
            p = wave_sum(p);
            if (lane == 0) red[1][tt][wave] = p;
        }
        __syncthreads();
        #pragma unroll
        for (int tt = 0; tt < TMAX; tt++) {
            float rs = 0.0f;
            #pragma unroll
            for (int w = 0; w < 8; w++) rs += red[1][tt][w];
            const float rstd = 1.0f / sqrtf(rs / K + a.eps);
            if (tt < T && tid < K4) {
                float o[4] = { (xr[tt].x - mean[tt]) * rstd, (xr[tt].y - mean[tt]) * rstd, (xr[tt].z - mean[tt]) * rstd, (xr[tt].w - mean[tt]) * rstd };
                o[0] = o[0]*lw.x; o[1] = o[1]*lw.y; o[2] = o[2]*lw.z; o[3] = o[3]*lw.w;
                o[0] = o[0]+lb.x; o[1] = o[1]+lb.y; o[2] = o[2]+lb.z; o[3] = o[3]+lb.w;
                dg_q8_0_store(o, tid*4, tt, nb, plo, phi, pdx, psx);
            }
        }
    } else {
        const int n16 = (int) (dg_planes_bytes(WT, K, Tall) >> 4);          // all images, as they lie in HBM
        for (int idx = tid; idx < n16; idx += 512) ((u32x4 *) smem)[idx] = ((const u32x4 *) a.xq)[idx];
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    constexpr int off = WT == MI355X_TYPE_Q5_0 ? 16 : (WT == MI355X_TYPE_Q4_0 ? 8 : 0);
    const size_t istride = dg_img_stride(WT, K);

    // T = 1: this lane's activation blocks never change
    uint4 ral[TMAX == 1 ? U : 1], rah[TMAX == 1 ? U : 1]; float rdx[TMAX == 1 ? U : 1]; int rsx[TMAX == 1 ? U : 1];
    if constexpr (TMAX == 1) {
        const uint4 * alo = (const uint4 *) plo, * ahi = (const uint4 *) phi;
        #pragma unroll
        for (int u = 0; u < U; u++) { const int g = j8 + 8*u; ral[u] = alo[g]; rah[u] = ahi[g]; rdx[u] = pdx[g]; rsx[u] = off * psx[g]; }
    }
    // this lane's destination columns (lane j8 == t finishes column t of every image)
    float * dcol[NG], * mcol[NG];
    #pragma unroll
    for (int gi = 0; gi < NG; gi++) {
        dcol[gi] = (float *) a.dstcol[gi*MI355X_IMG_COLS]; mcol[gi] = (float *) a.mircol[gi*MI355X_IMG_COLS];
        #pragma unroll
        for (int t = 1; t < TMAX; t++) { dcol[gi] = j8 == t ? (float *) a.dstcol[gi*MI355X_IMG_COLS + t] : dcol[gi]; mcol[gi] = j8 == t ? (float *) a.mircol[gi*MI355X_IMG_COLS + t] : mcol[gi]; }
    }

    for (; grp < ngroups; grp += 2*stride) {
      #pragma unroll
      for (int half = 0; half < 2; half++) {
        const int g_ = grp + half*stride;
        if (g_ >= ngroups) break;
        wblk<WT> * buf = half == 0 ? A : B;
        const int row = g_*8 + r8;
        // the row group's weights are unpacked once and meet every image (NG > 1: image after image, the compiler kept from interleaving them)
        uint32_t vlo[U][4], vhi[U][4]; float dw[U];
        if constexpr (NG > 1) {
            #pragma unroll
            for (int u = 0; u < U; u++) { wblk_unpack<WT>(buf[u], vlo[u], vhi[u]); dw[u] = h2f(buf[u].d); }
        }
        #pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            const int Ti = NG == 1 ? T : (Tall - gi*MI355X_IMG_COLS < MI355X_IMG_COLS ? Tall - gi*MI355X_IMG_COLS : MI355X_IMG_COLS);
            const uint4 * alo = (const uint4 *) (smem + gi*istride), * ahi = alo + (size_t) Ti*nb;
            const float * idx_ = (const float *) (ahi + (size_t) Ti*nb);
            const int * isx = (const int *) (idx_ + Ti*nb);
            float acc[TMAX];
            #pragma unroll
            for (int t = 0; t < TMAX; t++) acc[t] = 0.0f;
            #pragma unroll
            for (int u = 0; u < U; u++) {
                const int g = j8 + 8*u;
                if constexpr (NG == 1) { wblk_unpack<WT>(buf[u], vlo[u], vhi[u]); dw[u] = h2f(buf[u].d); }      // (one image: unpacked where it is used — registers)
                #pragma unroll
                for (int tt = 0; tt < TMAX; tt++) {
                    uint4 al, ah; float dxv; int sxo;
                    if constexpr (TMAX == 1) { al = ral[u]; ah = rah[u]; dxv = rdx[u]; sxo = rsx[u]; }
                    else { const int t = tt < Ti ? tt : Ti - 1; al = alo[(size_t) t*nb + g]; ah = ahi[(size_t) t*nb + g]; dxv = idx_[t*nb + g]; sxo = off ? off * isx[t*nb + g] : 0; }
                    int sum = 0;
                    sum = __builtin_amdgcn_sdot4((int) vlo[u][0], (int) al.x, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[u][1], (int) al.y, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[u][2], (int) al.z, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vlo[u][3], (int) al.w, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[u][0], (int) ah.x, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[u][1], (int) ah.y, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[u][2], (int) ah.z, sum, false);
                    sum = __builtin_amdgcn_sdot4((int) vhi[u][3], (int) ah.w, sum, false);
                    if (off) sum -= sxo;
                    acc[tt] = fmaf(dw[u] * dxv, (float) sum, acc[tt]);
                    if constexpr (TMAX > 2) __builtin_amdgcn_sched_barrier(0);      // (keeps the LDS reads of later columns from being hoisted: registers)
                }
                if constexpr (TMAX == 2) __builtin_amdgcn_sched_barrier(0);
            }
            #pragma unroll
            for (int t = 0; t < TMAX; t++) acc[t] = group_sum<8>(acc[t]);
            float v = acc[0];
            #pragma unroll
            for (int t = 1; t < TMAX; t++) v = (j8 == t) ? acc[t] : v;
            if (row < N && j8 < Ti) {
                dcol[gi][row] = v;
                if (mcol[gi]) mcol[gi][row] = v;
            }
            if constexpr (NG > 1) __builtin_amdgcn_sched_barrier(0);
        }
        load_group(buf, g_ + 2*stride);
      }
    }
}

template <int WT, int TMAX, int NG>
static int launch_vocab_u(mi355x_ctx * ctx, const VArgs & k, int U, dim3 grid, uint32_t lds, double bytes, double flops) {
    switch (U) {
        case 2: return emit(ctx, "vocab", k_vocab<WT, TMAX, 2, NG>, grid, dim3(512), lds, k, bytes, flops);
        case 3: return emit(ctx, "vocab", k_vocab<WT, TMAX, 3, NG>, grid, dim3(512), lds, k, bytes, flops);
        case 4: return emit(ctx, "vocab", k_vocab<WT, TMAX, 4, NG>, grid, dim3(512), lds, k, bytes, flops);
        case 5: return emit(ctx, "vocab", k_vocab<WT, TMAX, 5, NG>, grid, dim3(512), lds, k, bytes, flops);
    }
    return MI355X_E_UNSUPPORTED;
}
template <int WT>
static int launch_vocab(mi355x_ctx * ctx, const VArgs & k, int U, dim3 grid, uint32_t lds, double bytes, double flops) {
    if (k.T == 1) return launch_vocab_u<WT, 1, 1>(ctx, k, U, grid, lds, bytes, flops);
    if (k.T == 2) return launch_vocab_u<WT, 2, 1>(ctx, k, U, grid, lds, bytes, flops);
    if (k.T <= 4) return launch_vocab_u<WT, 4, 1>(ctx, k, U, grid, lds, bytes, flops);
    if (k.T <= 8) return launch_vocab_u<WT, 8, 1>(ctx, k, U, grid, lds, bytes, flops);
    if (k.T <= 16) return launch_vocab_u<WT, 8, 2>(ctx, k, U, grid, lds, bytes, flops);
    if (k.T <= 24) return launch_vocab_u<WT, 8, 3>(ctx, k, U, grid, lds, bytes, flops);
    return launch_vocab_u<WT, 8, 4>(ctx, k, U, grid, lds, bytes, flops);
}

// the vocabulary projection: MI355X_E_UNSUPPORTED = not this shape (the caller goes on to k_gemv8)
int mi355x_vocab(mi355x_ctx * ctx, const mi355x_gemv_desc * d) {
    if (d->nseg != 1 || d->T < 1 || d->T > MI355X_MAX_COLS || d->attn_part_o) return MI355X_E_UNSUPPORTED;
    const mi355x_gemv_seg & g = d->seg[0];
    const int wt = g.wtype, K = d->K, T = d->T, N = g.N;
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0) return MI355X_E_UNSUPPORTED;
    if (N <= 8192 || K <= 0 || K % 256 || K > 2048 || K / 256 < 2 || K / 256 > 5) return MI355X_E_UNSUPPORTED;
    if (g.ep.bias || g.ep.has_scale || g.ep.gelu || g.ep.residual || g.dst_type != MI355X_TYPE_F32 || ((uintptr_t) g.w % 16)) return MI355X_E_UNSUPPORTED;
    const bool planes = d->x_planes != nullptr;
    if (planes ? (d->x != nullptr || d->has_norm || ((uintptr_t) d->x_planes % 16)) : (!d->x || !d->has_norm)) return MI355X_E_UNSUPPORTED;
    VArgs k; memset(&k, 0, sizeof(k));
    k.w = g.w; k.nbt = (int64_t) N * (K / 32); k.N = N; k.K = K; k.T = T;
    if (planes) k.xq = d->x_planes;
    else {
        if (!d->ln_w || !d->ln_b || (((uintptr_t) d->x | (uintptr_t) d->ln_w | (uintptr_t) d->ln_b) % 16) || (d->x_nb1 % 16)) return MI355X_E_UNSUPPORTED;
        k.has_norm = 1; k.x = d->x; k.x_nb1 = d->x_nb1; k.ln_w = d->ln_w; k.ln_b = d->ln_b; k.eps = d->eps;
    }
    const bool cols = planes && d->cols;
    bool mirror = d->cols && d->cols->mirror[0];
    for (int t = 0; t < MI355X_MAX_COLS; t++) {
        const int tc = t < T ? t : T - 1;
        k.dstcol[t] = cols ? d->cols->dst[0][tc] : (g.dst ? (char *) g.dst + (int64_t) tc*g.dst_nb1 : nullptr);
        k.mircol[t] = mirror ? d->cols->mirror[tc] : nullptr;
        if (!k.dstcol[t] || ((uintptr_t) k.dstcol[t] % 4) || (mirror && !k.mircol[t])) return MI355X_E_UNSUPPORTED;
    }
    if (T > MI355X_IMG_COLS && !planes) return MI355X_E_UNSUPPORTED;          // more than one image: prepared planes only
    const size_t lds = dg_planes_bytes(wt, K, T) + 16;
    if (lds > 64 * 1024) return MI355X_E_UNSUPPORTED;
    // 8-wave workgroups, every wave walking 4 row groups with two of them in flight (measured, large-v3 Q5_0, HBM-cold, rocprofv3:
    // 4 groups per wave = 203 workgroups 11.1 us, one workgroup per CU 11.6, 2 groups per wave 12.3, 1 per wave 14.1, 8 per wave 15.2;
    // k_gemv8 17.9; a read-once stream of the same bytes 9.0 — profiles/r03_vocab_kernel.txt).
    const int ngroups = (N + 7) / 8;
    int nwg = ((ngroups + 3) / 4 + 7) / 8;
    if (nwg > (ngroups + 7) / 8) nwg = (ngroups + 7) / 8;
    const dim3 grid(nwg);
    const double bytes = (double) mi355x_type_row_bytes(wt, K) * N + (double) K*T*4 + (double) N*T*4;
    const double flops = 2.0 * N * K * T;
    int rc;
    switch (wt) {
        case MI355X_TYPE_Q4_0: rc = launch_vocab<MI355X_TYPE_Q4_0>(ctx, k, K / 256, grid, (uint32_t) lds, bytes, flops); break;
        case MI355X_TYPE_Q5_0: rc = launch_vocab<MI355X_TYPE_Q5_0>(ctx, k, K / 256, grid, (uint32_t) lds, bytes, flops); break;
        default:               rc = launch_vocab<MI355X_TYPE_Q8_0>(ctx, k, K / 256, grid, (uint32_t) lds, bytes, flops); break;
    }
    if (rc == 0 && mirror) ctx->last_mirrored = 1;
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// cross-state batches: attention and step head for S single-token states in one launch
// ---------------------------------------------------------------------------------------------------
struct FDMState { const char * q; const char * k; const char * v; const char * m; int n_kv; int pad; };
struct FDMArgs {
    FDMState st[MI355X_MAX_COLS];
    int64_t q_nb2, k_nb1, k_nb2, v_nb1, v_nb2;
    float scale; int S, H, rk2, rv2, nparts;
    float * part_o; float * part_ml;
};

// k_fattn_dec<1> (decode.hip) with the state selected by blockIdx.z: 128 keys per workgroup (4 waves x 32 keys), record
// ((h*S + s)*nparts + p).  A chunk beyond the state's keys yields the empty record (m = -1e30, l = 0, o = 0).
__global__ void __launch_bounds__(256) k_fattn_dec_multi(const FDMArgs a) {
    __shared__ __attribute__((aligned(16))) float wo[4][64];
    __shared__ float wml[4][2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kg = lane >> 3, dc = lane & 7;
    const int hq = blockIdx.y, hk = hq / a.rk2, hv = hq / a.rv2;
    const int p = blockIdx.x, si = blockIdx.z;
    const FDMState & st = a.st[si];
    const int n_kv = st.n_kv;
    const bool has_mask = st.m != nullptr;
    const int kbeg = p*128 + wave*32;
    const char * kbase = st.k + (int64_t) hk*a.k_nb2 + dc*16;
    const char * vbase = st.v + (int64_t) hv*a.v_nb2 + dc*16;
    uint4 kr[4], vr[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int key = kbeg + kg + 8*i, kc = key < n_kv ? key : n_kv - 1;
        kr[i] = *(const uint4 *) (kbase + (int64_t) kc*a.k_nb1);
        vr[i] = *(const uint4 *) (vbase + (int64_t) kc*a.v_nb1);
    }
    float qf[8];
    {
        const float * qp = (const float *) (st.q + (int64_t) hq*a.q_nb2) + dc*8;
        const float4 q0 = *(const float4 *) qp, q1 = *(const float4 *) (qp + 4);
        qf[0] = round_f16(q0.x); qf[1] = round_f16(q0.y); qf[2] = round_f16(q0.z); qf[3] = round_f16(q0.w);
        qf[4] = round_f16(q1.x); qf[5] = round_f16(q1.y); qf[6] = round_f16(q1.z); qf[7] = round_f16(q1.w);
    }
    uint16_t mkh[4];
    const char * mbase = has_mask ? st.m : st.k;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int key = kbeg + kg + 8*i, kc = key < n_kv ? key : n_kv - 1;
        mkh[i] = *(const uint16_t *) (mbase + (int64_t) kc*2);
    }
    float sc[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int key = kbeg + kg + 8*i;
        const uint32_t w[4] = { kr[i].x, kr[i].y, kr[i].z, kr[i].w };
        float kf[8];
        #pragma unroll
        for (int e = 0; e < 4; e++) { kf[2*e] = h2f((uint16_t) (w[e] & 0xFFFF)); kf[2*e+1] = h2f((uint16_t) (w[e] >> 16)); }
        float s = 0.0f;
        #pragma unroll
        for (int e = 0; e < 8; e++) s = fmaf(kf[e], qf[e], s);
        s = group_sum<8>(s);
        const float x = s * a.scale + (has_mask ? h2f(mkh[i]) : 0.0f);
        sc[i] = key < n_kv ? x : -INFINITY;
    }
    {
        float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        m = stride8_max(m);
        m = fmaxf(m, -1e30f);
        float l = 0.0f, o[8];
        #pragma unroll
        for (int e = 0; e < 8; e++) o[e] = 0.0f;
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const float pk = __expf(sc[i] - m);
            l += pk;
            const uint32_t w[4] = { vr[i].x, vr[i].y, vr[i].z, vr[i].w };
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                o[2*e]   = fmaf(pk, h2f((uint16_t) (w[e] & 0xFFFF)), o[2*e]);
                o[2*e+1] = fmaf(pk, h2f((uint16_t) (w[e] >> 16)),    o[2*e+1]);
            }
        }
        l = stride8_sum(l);
        #pragma unroll
        for (int e = 0; e < 8; e++) o[e] = stride8_sum(o[e]);
        if (kg == 0) {
            *(float4 *) &wo[wave][dc*8]     = make_float4(o[0], o[1], o[2], o[3]);
            *(float4 *) &wo[wave][dc*8 + 4] = make_float4(o[4], o[5], o[6], o[7]);
            if (dc == 0) { wml[wave][0] = m; wml[wave][1] = l; }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int d = tid;
        const float m0 = wml[0][0], m1 = wml[1][0], m2 = wml[2][0], m3 = wml[3][0];
        const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float w0 = __expf(m0 - M), w1 = __expf(m1 - M), w2 = __expf(m2 - M), w3 = __expf(m3 - M);
        const float o = fmaf(w3, wo[3][d], fmaf(w2, wo[2][d], fmaf(w1, wo[1][d], w0 * wo[0][d])));
        const int64_t rec = ((int64_t) hq*a.S + si) * a.nparts + p;
        a.part_o[rec*64 + d] = o;
        if (d == 0) {
            a.part_ml[rec*2]     = M;
            a.part_ml[rec*2 + 1] = fmaf(w3, wml[3][1], fmaf(w2, wml[2][1], fmaf(w1, wml[1][1], w0 * wml[0][1])));
        }
    }
}

extern "C" int mi355x_flash_attn_partial_multi(mi355x_ctx * ctx, int S, const mi355x_attn_state * st, const mi355x_tensor * q, const mi355x_tensor * k,
                                               const mi355x_tensor * v, float scale, mi355x_attn_partials * out) {
    if (S < 1 || S > MI355X_MAX_COLS) return MI355X_E_UNSUPPORTED;
    if (q->type != MI355X_TYPE_F32 || k->type != MI355X_TYPE_F16 || v->type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
    if (q->ne[0] != 64 || k->ne[0] != 64 || v->ne[0] != 64 || q->ne[1] != 1 || q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return MI355X_E_UNSUPPORTED;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2) return MI355X_E_UNSUPPORTED;
    const int H = (int) q->ne[2];
    if (H < 1 || k->ne[2] <= 0 || H % k->ne[2] || v->ne[2] <= 0 || H % v->ne[2]) return MI355X_E_UNSUPPORTED;
    if ((q->nb[2] | k->nb[1] | k->nb[2] | v->nb[1] | v->nb[2]) % 16) return MI355X_E_UNSUPPORTED;
    FDMArgs a; memset(&a, 0, sizeof(a));
    int nparts = 1;
    for (int s = 0; s < S; s++) {
        if (!st[s].q || !st[s].k || !st[s].v || st[s].n_kv < 1) return MI355X_E_UNSUPPORTED;
        if (((uintptr_t) st[s].q | (uintptr_t) st[s].k | (uintptr_t) st[s].v) % 16 || ((uintptr_t) st[s].mask % 2)) return MI355X_E_UNSUPPORTED;
        a.st[s].q = (const char *) st[s].q; a.st[s].k = (const char *) st[s].k; a.st[s].v = (const char *) st[s].v; a.st[s].m = (const char *) st[s].mask;
        a.st[s].n_kv = st[s].n_kv;
        const int np = (st[s].n_kv + 127) / 128;
        if (np > nparts) nparts = np;
    }
    a.q_nb2 = q->nb[2]; a.k_nb1 = k->nb[1]; a.k_nb2 = k->nb[2]; a.v_nb1 = v->nb[1]; a.v_nb2 = v->nb[2];
    a.scale = scale; a.S = S; a.H = H; a.rk2 = (int) (H / k->ne[2]); a.rv2 = (int) (H / v->ne[2]); a.nparts = nparts;
    mi355x_scratch_reset(ctx);
    const size_t nrec = (size_t) H * S * nparts;
    a.part_o  = (float *) mi355x_scratch_alloc(ctx, nrec * 64 * 4);
    a.part_ml = (float *) mi355x_scratch_alloc(ctx, nrec * 2 * 4);
    if (!a.part_o || !a.part_ml) return (int) hipErrorOutOfMemory;
    double bytes = 0;
    for (int s = 0; s < S; s++) bytes += 2.0 * st[s].n_kv * 64 * 2 * H;
    bytes += (double) S*H*64*4 + (double) nrec*66*4;
    double flops = 0;
    for (int s = 0; s < S; s++) flops += 4.0 * (double) st[s].n_kv * 64 * H;
    const int rc = emit(ctx, "fattn_dec_multi", k_fattn_dec_multi, dim3(nparts, H, S), dim3(256), 0, a, bytes, flops);
    if (rc) return rc;
    out->part_o = a.part_o; out->part_ml = a.part_ml; out->nparts = nparts; out->T = S; out->H = H;
    return 0;
}

// Attention of a decoder step (n_kv <= 512 * ROUNDS: self-attention, and the 1500 keys of cross-attention in three rounds) COMPLETE
// in one launch per layer: one 16-wave workgroup per (head, column).  In round r wave w takes keys [32 (16 r + w), + 32) exactly as a
// wave of k_fattn_dec does (the next round's K / V rows are requested before this round's are used), the four waves of a 128-key chunk
// are merged into that chunk's record in LDS, at the end the records are combined and the head's 64 outputs leave as two Q8_0 blocks of
// the output projection's activation planes.
// Arithmetic = k_fattn_dec -> (records) -> k_act_prepare MODE 2 -> dg_q8_0_store, statement for statement: bit-identical to the three
// launches it replaces (attention, combine, quantize), one or two dependent launches less per layer.
template <int ROUNDS>
__global__ void __launch_bounds__(1024) k_fattn_self_q(const FDMArgs a) {
    __shared__ __attribute__((aligned(16))) float wo[16][64];
    __shared__ float wml[16][2];
    __shared__ __attribute__((aligned(16))) float ro[ROUNDS*4][64];
    __shared__ float rml[ROUNDS*4][2];
    __shared__ __attribute__((aligned(16))) float xo[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kg = lane >> 3, dc = lane & 7;
    const int hq = blockIdx.x, hk = hq / a.rk2, hv = hq / a.rv2;
    const int si = blockIdx.y;
    const FDMState & st = a.st[si];
    const int n_kv = st.n_kv;
    const bool has_mask = st.m != nullptr;
    const char * kbase = st.k + (int64_t) hk*a.k_nb2 + dc*16;
    const char * vbase = st.v + (int64_t) hv*a.v_nb2 + dc*16;
    const char * mbase = has_mask ? st.m : st.k;
    uint4 kr[2][4], vr[2][4];
    uint16_t mkh[2][4];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int key = wave*32 + kg + 8*i, kc = key < n_kv ? key : n_kv - 1;
        kr[0][i] = *(const uint4 *) (kbase + (int64_t) kc*a.k_nb1);
        vr[0][i] = *(const uint4 *) (vbase + (int64_t) kc*a.v_nb1);
    }
    float qf[8];
    {
        const float * qp = (const float *) (st.q + (int64_t) hq*a.q_nb2) + dc*8;
        const float4 q0 = *(const float4 *) qp, q1 = *(const float4 *) (qp + 4);
        qf[0] = round_f16(q0.x); qf[1] = round_f16(q0.y); qf[2] = round_f16(q0.z); qf[3] = round_f16(q0.w);
        qf[4] = round_f16(q1.x); qf[5] = round_f16(q1.y); qf[6] = round_f16(q1.z); qf[7] = round_f16(q1.w);
    }
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int key = wave*32 + kg + 8*i, kc = key < n_kv ? key : n_kv - 1;
        mkh[0][i] = *(const uint16_t *) (mbase + (int64_t) kc*2);
    }
    #pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        constexpr int dummy = 0; (void) dummy;
        const int cur = r & 1, nxt = cur ^ 1;
        const int kbeg = (r*16 + wave)*32;
        if (r + 1 < ROUNDS) {
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kbeg + 512 + kg + 8*i, kc = key < n_kv ? key : n_kv - 1;
                kr[nxt][i] = *(const uint4 *) (kbase + (int64_t) kc*a.k_nb1);
                vr[nxt][i] = *(const uint4 *) (vbase + (int64_t) kc*a.v_nb1);
                mkh[nxt][i] = *(const uint16_t *) (mbase + (int64_t) kc*2);
            }
        }
        float sc[4];
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const int key = kbeg + kg + 8*i;
            const uint32_t w[4] = { kr[cur][i].x, kr[cur][i].y, kr[cur][i].z, kr[cur][i].w };
            float kf[8];
            #pragma unroll
            for (int e = 0; e < 4; e++) { kf[2*e] = h2f((uint16_t) (w[e] & 0xFFFF)); kf[2*e+1] = h2f((uint16_t) (w[e] >> 16)); }
            float s = 0.0f;
            #pragma unroll
            for (int e = 0; e < 8; e++) s = fmaf(kf[e], qf[e], s);
            s = group_sum<8>(s);
            const float x = s * a.scale + (has_mask ? h2f(mkh[cur][i]) : 0.0f);
            sc[i] = key < n_kv ? x : -INFINITY;
        }
        {
            float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            m = stride8_max(m);
            m = fmaxf(m, -1e30f);
            float l = 0.0f, o[8];
            #pragma unroll
            for (int e = 0; e < 8; e++) o[e] = 0.0f;
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const float pk = __expf(sc[i] - m);
                l += pk;
                const uint32_t w[4] = { vr[cur][i].x, vr[cur][i].y, vr[cur][i].z, vr[cur][i].w };
                #pragma unroll
                for (int e = 0; e < 4; e++) {
                    o[2*e]   = fmaf(pk, h2f((uint16_t) (w[e] & 0xFFFF)), o[2*e]);
                    o[2*e+1] = fmaf(pk, h2f((uint16_t) (w[e] >> 16)),    o[2*e+1]);
                }
            }
            l = stride8_sum(l);
            #pragma unroll
            for (int e = 0; e < 8; e++) o[e] = stride8_sum(o[e]);
            if (kg == 0) {
                *(float4 *) &wo[wave][dc*8]     = make_float4(o[0], o[1], o[2], o[3]);
                *(float4 *) &wo[wave][dc*8 + 4] = make_float4(o[4], o[5], o[6], o[7]);
                if (dc == 0) { wml[wave][0] = m; wml[wave][1] = l; }
            }
        }
        __syncthreads();
        if (tid < 256) {
            // chunk p = tid / 64 of this round: the merge of its four waves, as at the end of k_fattn_dec
            const int p = tid >> 6, d = tid & 63, w0i = p*4;
            const float m0 = wml[w0i][0], m1 = wml[w0i + 1][0], m2 = wml[w0i + 2][0], m3 = wml[w0i + 3][0];
            const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
            const float w0 = __expf(m0 - M), w1 = __expf(m1 - M), w2 = __expf(m2 - M), w3 = __expf(m3 - M);
            ro[r*4 + p][d] = fmaf(w3, wo[w0i + 3][d], fmaf(w2, wo[w0i + 2][d], fmaf(w1, wo[w0i + 1][d], w0 * wo[w0i][d])));
            if (d == 0) {
                rml[r*4 + p][0] = M;
                rml[r*4 + p][1] = fmaf(w3, wml[w0i + 3][1], fmaf(w2, wml[w0i + 2][1], fmaf(w1, wml[w0i + 1][1], w0 * wml[w0i][1])));
            }
        }
        __syncthreads();                                          // wo / wml are rewritten by the next round; ro / rml are read below
    }
    const int nparts = (n_kv + 127) >> 7;                       // <= 4 * ROUNDS
    if (tid < 64) {
        // combine of the chunk records, as k_act_prepare MODE 2 / k_gemv_row MODE 2 do it
        float M = -1e30f, L = 0.0f, o = 0.0f;
        for (int p = 0; p < nparts; p++) M = fmaxf(M, rml[p][0]);
        for (int p = 0; p < nparts; p++) {
            const float w = __expf(rml[p][0] - M);
            L = fmaf(w, rml[p][1], L);
            o = fmaf(w, ro[p][tid], o);
        }
        const float inv = L == 0.0f ? 0.0f : 1.0f / L;
        xo[tid] = o*inv;
    }
    __syncthreads();
    if (tid < 16) {
        // the head's 64 values = blocks 2h and 2h + 1 of column si: lane j quantizes values 4j .. 4j + 3 (8 lanes per block)
        const float4 x4 = *(const float4 *) &xo[tid*4];
        const float xv[4] = { x4.x, x4.y, x4.z, x4.w };
        // (column si = column si % 8 of image si / 8)
        const int K = a.H*64, nb = K >> 5;
        const int ti = si & (MI355X_IMG_COLS - 1), T = a.S - (si - ti) < MI355X_IMG_COLS ? a.S - (si - ti) : MI355X_IMG_COLS;
        uint32_t * lo = (uint32_t *) ((char *) a.part_o + (size_t) (si >> 3) * dg_img_stride(MI355X_TYPE_Q8_0, K)), * hi = lo + (size_t) T*nb*4;
        float * dx = (float *) (hi + (size_t) T*nb*4);
        int * sx = (int *) (dx + T*nb);
        dg_q8_0_store(xv, hq*64 + tid*4, ti, nb, lo, hi, dx, sx);
    }
}

// decoder attention for T columns (own q / K / V / mask row / key count each; up to 1536 keys: self- and cross-attention) straight into the
// Q8_0 planes of the output projection
extern "C" int mi355x_flash_attn_planes(mi355x_ctx * ctx, int T, const mi355x_attn_state * st, const mi355x_tensor * q, const mi355x_tensor * k,
                                        const mi355x_tensor * v, float scale, void * planes) {
    if (T < 1 || T > MI355X_MAX_COLS || !planes || ((uintptr_t) planes % 16)) return MI355X_E_UNSUPPORTED;
    if (q->type != MI355X_TYPE_F32 || k->type != MI355X_TYPE_F16 || v->type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
    if (q->ne[0] != 64 || k->ne[0] != 64 || v->ne[0] != 64 || q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return MI355X_E_UNSUPPORTED;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2) return MI355X_E_UNSUPPORTED;
    const int H = (int) q->ne[2];
    if (H < 1 || H*64 > 2048 || k->ne[2] <= 0 || H % k->ne[2] || v->ne[2] <= 0 || H % v->ne[2]) return MI355X_E_UNSUPPORTED;
    if ((q->nb[2] | k->nb[1] | k->nb[2] | v->nb[1] | v->nb[2]) % 16) return MI355X_E_UNSUPPORTED;
    FDMArgs a; memset(&a, 0, sizeof(a));
    double bytes = 0, flops = 0;
    int max_kv = 0;
    for (int s = 0; s < T; s++) {
        if (!st[s].q || !st[s].k || !st[s].v || st[s].n_kv < 1 || st[s].n_kv > 1536) return MI355X_E_UNSUPPORTED;
        if (st[s].n_kv > max_kv) max_kv = st[s].n_kv;
        if (((uintptr_t) st[s].q | (uintptr_t) st[s].k | (uintptr_t) st[s].v) % 16 || ((uintptr_t) st[s].mask % 2)) return MI355X_E_UNSUPPORTED;
        a.st[s].q = (const char *) st[s].q; a.st[s].k = (const char *) st[s].k; a.st[s].v = (const char *) st[s].v; a.st[s].m = (const char *) st[s].mask;
        a.st[s].n_kv = st[s].n_kv;
        bytes += 2.0 * st[s].n_kv * 64 * 2 * H + (double) H*64*4; flops += 4.0 * (double) st[s].n_kv * 64 * H;
    }
    a.q_nb2 = q->nb[2]; a.k_nb1 = k->nb[1]; a.k_nb2 = k->nb[2]; a.v_nb1 = v->nb[1]; a.v_nb2 = v->nb[2];
    a.scale = scale; a.S = T; a.H = H; a.rk2 = (int) (H / k->ne[2]); a.rv2 = (int) (H / v->ne[2]); a.nparts = 0;
    a.part_o = (float *) planes;            // (the planes travel in the record pointer's slot)
    bytes += (double) T*H*64*1.25;
    if (max_kv <= 512)  return emit(ctx, "fattn_self_q", k_fattn_self_q<1>, dim3(H, T), dim3(1024), 0, a, bytes, flops);
    if (max_kv <= 1024) return emit(ctx, "fattn_self_q", k_fattn_self_q<2>, dim3(H, T), dim3(1024), 0, a, bytes, flops);
    return emit(ctx, "fattn_self_q", k_fattn_self_q<3>, dim3(H, T), dim3(1024), 0, a, bytes, flops);
}

struct HeadArgs {
    mi355x_head_state st[MI355X_MAX_COLS];
    const char * te; int64_t te_nbt, te_nb1; int te_rows; int te_type;
    const char * pe; int64_t pe_nb1; int pe_rows; int ne0; int S;
};
// blocks [0, S): embedding of state s; blocks [S, 2S): mask cast of state s - S
template <int TYPE>
__global__ void __launch_bounds__(256) k_decode_head_multi(const HeadArgs a) {
    const int b = blockIdx.x;
    if (b >= a.S) {
        const mi355x_head_state & st = a.st[b - a.S];
        if (!st.mask_f32) return;
        for (int i = threadIdx.x; i < st.n_mask; i += 256) ((uint16_t *) st.mask_f16)[i] = f2h(st.mask_f32[i]);
        return;
    }
    const mi355x_head_state & st = a.st[b];
    if (!st.tok) return;
    const int32_t row = *st.tok, ar = *st.pos;
    float * dst = st.dst;
    const int ne0 = a.ne0;
    if (row < 0 || row >= a.te_rows || ar < 0 || ar >= a.pe_rows) {
        // an id or position outside the tables (the reference's get_rows asserts, ggml-cpu/ops.cpp:4850-5017): the state's embedding becomes NaN,
        // so that its logits are visibly wrong instead of the previous step's stale vector; other states are unaffected
        for (int i = threadIdx.x; i < ne0; i += 256) dst[i] = __uint_as_float(0x7FC00000u);
        return;
    }
    const float * addrow = (const float *) (a.pe + (int64_t) ar*a.pe_nb1);
    if constexpr (TYPE == MI355X_TYPE_F32 || TYPE == MI355X_TYPE_F16) {
        const char * src = a.te + (int64_t) row*a.te_nb1;
        for (int i = threadIdx.x; i < ne0; i += 256) {
            const float v = TYPE == MI355X_TYPE_F32 ? ((const float *) src)[i] : h2f(((const uint16_t *) src)[i]);
            dst[i] = v + addrow[i];
        }
    } else {
        const qplanes<TYPE> p(a.te, a.te_nbt);
        const int nb32 = ne0 / 32;
        for (int g = threadIdx.x; g < nb32; g += 256) {
            float v[32];
            dequant_block32<TYPE>(p, (int64_t) row * nb32 + g, v);
            #pragma unroll
            for (int j = 0; j < 32; j += 4) { const float4 q = *(const float4 *) (addrow + g*32 + j); v[j] += q.x; v[j+1] += q.y; v[j+2] += q.z; v[j+3] += q.w; }
            #pragma unroll
            for (int j = 0; j < 32; j += 4) *(float4 *) (dst + g*32 + j) = make_float4(v[j], v[j+1], v[j+2], v[j+3]);
        }
    }
}

extern "C" int mi355x_decode_head_multi(mi355x_ctx * ctx, int S, const mi355x_head_state * st, const mi355x_tensor * te, const mi355x_tensor * pe) {
    if (S < 1 || S > MI355X_MAX_COLS) return MI355X_E_UNSUPPORTED;
    bool any_tok = false;
    for (int s = 0; s < S; s++) any_tok = any_tok || st[s].tok != nullptr;
    if (!any_tok) {                  // casts only: the embedding tables are not needed
        HeadArgs a; memset(&a, 0, sizeof(a));
        for (int s = 0; s < S; s++) { if (st[s].mask_f32 && (!st[s].mask_f16 || st[s].n_mask < 1)) return MI355X_E_UNSUPPORTED; a.st[s] = st[s]; }
        a.S = S;
        return emit(ctx, "decode_head_multi", k_decode_head_multi<MI355X_TYPE_F32>, dim3(2 * S), dim3(256), 0, a, 0, 0);
    }
    if (!te || !pe) return MI355X_E_UNSUPPORTED;
    if (pe->type != MI355X_TYPE_F32 || pe->nb[0] != 4 || pe->ne[0] != te->ne[0] || pe->ne[2] != 1 || pe->ne[3] != 1 || te->ne[2] != 1 || te->ne[3] != 1 ||
        ((uintptr_t) pe->data % 16) || (pe->nb[1] % 16)) return MI355X_E_UNSUPPORTED;
    HeadArgs a; memset(&a, 0, sizeof(a));
    for (int s = 0; s < S; s++) {
        if (st[s].tok && (!st[s].pos || !st[s].dst || ((uintptr_t) st[s].dst % 16))) return MI355X_E_UNSUPPORTED;     // tok == NULL: no embedding for this state
        if (st[s].mask_f32 && (!st[s].mask_f16 || st[s].n_mask < 1)) return MI355X_E_UNSUPPORTED;
        a.st[s] = st[s];
    }
    a.te = (const char *) te->data; a.te_nb1 = te->nb[1]; a.te_rows = (int) te->ne[1]; a.te_type = te->type;
    a.pe = (const char *) pe->data; a.pe_nb1 = pe->nb[1]; a.pe_rows = (int) pe->ne[1]; a.ne0 = (int) te->ne[0]; a.S = S;
    if (mi355x_type_is_quantized(te->type)) {
        if (!t_is_contiguous(te) || te->ne[0] % 32) return MI355X_E_UNSUPPORTED;
        a.te_nbt = t_nelements(te) / type_block(te->type);
    }
    const double bytes = (double) S * (mi355x_type_row_bytes(te->type, te->ne[0]) + te->ne[0]*8.0);
    const dim3 g(2 * S), b(256);
    switch (te->type) {
        case MI355X_TYPE_F32:  if (te->nb[0] != 4) return MI355X_E_UNSUPPORTED; return emit(ctx, "decode_head_multi", k_decode_head_multi<MI355X_TYPE_F32>,  g, b, 0, a, bytes, 0);
        case MI355X_TYPE_F16:  if (te->nb[0] != 2) return MI355X_E_UNSUPPORTED; return emit(ctx, "decode_head_multi", k_decode_head_multi<MI355X_TYPE_F16>,  g, b, 0, a, bytes, 0);
        case MI355X_TYPE_Q4_0: return emit(ctx, "decode_head_multi", k_decode_head_multi<MI355X_TYPE_Q4_0>, g, b, 0, a, bytes, 0);
        case MI355X_TYPE_Q5_0: return emit(ctx, "decode_head_multi", k_decode_head_multi<MI355X_TYPE_Q5_0>, g, b, 0, a, bytes, 0);
        case MI355X_TYPE_Q8_0: return emit(ctx, "decode_head_multi", k_decode_head_multi<MI355X_TYPE_Q8_0>, g, b, 0, a, bytes, 0);
        case MI355X_TYPE_Q4_K: return emit(ctx, "decode_head_multi", k_decode_head_multi<MI355X_TYPE_Q4_K>, g, b, 0, a, bytes, 0);
        default: return MI355X_E_UNSUPPORTED;
    }
}
