// Decoder-step mat-vec (T <= 8 columns): the HBM-bandwidth-bound half of the hot path.
//
// Reference arithmetic being reproduced (ggml-cpu/ggml-cpu.c:1254-1452): src1 rows are first quantized to the
// weight type's vec_dot_type — Q8_0 blocks (d = amax/127 stored as f16, q = round-to-nearest-even(x*127/amax),
// arch/x86/quants.c:302-398) for Q4_0/Q5_0/Q8_0 weights, Q8_K (ggml-quants.c:2768-2805) for Q4_K, f16 for F16
// weights — then integer dot products per block, scaled by d_w*d_x and accumulated in f32
// (vec_dot q5_0: ggml-cpu/quants.c:365-406, q4_0 :225-259, q8_0 :451-479, q4_K :696-769).
//
// MI355X mapping: one wavefront per weight row (rows_per_wave rows in sequence), lanes stride over the row's
// blocks with aligned 16-byte loads from the planar layout, v_dot4_i32_i8 on bytes unpacked in registers, the
// quantized activations live in LDS (two 16-byte planes per block => conflict-free ds_read_b128), DPP/shuffle
// wave reduction, fused LayerNorm prologue and bias/scale/GELU/residual/f16 epilogue.  No LDS staging of
// weights: each weight byte is used exactly once.
#include "common.h"
#include <stdlib.h>

struct GemvSeg {
    const void *  w;  int64_t nbt;  int N;  int has_scale;
    const float * bias; float scale; int gelu;
    const float * residual; int64_t res_nb1;
    void * dst; int64_t dst_nb1; int dst_f16; int pad;
};
struct GemvArgs {
    const float * x; int64_t x_nb1; int K; int has_norm; float eps; int nseg;
    const float * ln_w; const float * ln_b;
    int rows_per_wave; int row_start[4];
    GemvSeg seg[3];
    const uint16_t * gelu_tab;
};

// ---- LDS layout helpers ---------------------------------------------------------------------------
// Q8_0 family: lo[T][nb] uint4 | hi[T][nb] uint4 | dx[T][nb] f32 | sx[T][nb] i32     (nb = K/32)
// Q8_K family: pl[4][T][nch] uint4 (nch = K/64) | dx[T][nsb] f32 | bs[T][nsb*8] i32   (nsb = K/256)
// F16        : xh[T][K] f16
static inline size_t gemv_lds_bytes(int wtype, int K, int T) {
    switch (wtype) {
        case MI355X_TYPE_Q4_0: case MI355X_TYPE_Q5_0: case MI355X_TYPE_Q8_0: return (size_t) T * (K/32) * 40 + 256;
        case MI355X_TYPE_Q4_K: return (size_t) T * ((K/64)*64 + (K/256)*4 + (K/32)*4) + 256;
        case MI355X_TYPE_F16:  return (size_t) T * K * 2 + 256;
        default: return 0;
    }
}

__device__ __forceinline__ void block_reduce_T(float * part, int T, float * red /* [4][8] */, int wave, int lane) {
    for (int t = 0; t < T; t++) { const float s = wave_sum(part[t]); if (lane == 0) red[wave*8 + t] = s; }
    __syncthreads();
    for (int t = 0; t < T; t++) part[t] = red[t] + red[8 + t] + red[16 + t] + red[24 + t];
}

// quantize 4 consecutive values (one lane of an 8-lane group = one 32-block) to Q8_0 and store to LDS
__device__ __forceinline__ void q8_0_store(const float v[4], int e, int t, int nb, uint32_t * lo, uint32_t * hi, float * dx, int * sx) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const float d  = amax / 127.0f;
    const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
    const int q0 = (int) rintf(v[0]*id), q1 = (int) rintf(v[1]*id), q2 = (int) rintf(v[2]*id), q3 = (int) rintf(v[3]*id);
    int s = q0 + q1 + q2 + q3;
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const uint32_t packed = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
    const int b = e >> 5, w = (e & 31) >> 2;
    uint32_t * plane = w < 4 ? lo : hi;
    plane[((size_t) t*nb + b)*4 + (w & 3)] = packed;
    if (w == 0) { dx[t*nb + b] = round_f16(d); sx[t*nb + b] = s; }
}

// quantize 4 consecutive values (one lane of a wave = one 256-block) to Q8_K and store to LDS
__device__ __forceinline__ void q8_K_store(const float v[4], int e, int t, int K, int T, uint32_t * pl, float * dx, int * bs) {
    float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
    const float amax = fmaxf(mx, -mn);
    const float maxv = (mx >= -mn) ? mx : mn;          // value with the largest magnitude (sign kept)
    const int nch = K >> 6, nsb = K >> 8;
    int q[4] = {0, 0, 0, 0};
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.0f / maxv;
        #pragma unroll
        for (int i = 0; i < 4; i++) { int r = (int) rintf(iscale * v[i]); q[i] = r < 127 ? r : 127; }
        d = 1.0f / iscale;
    }
    int s = q[0] + q[1] + q[2] + q[3];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);      // sum over 32 elements
    const uint32_t packed = (uint32_t) (q[0] & 0xFF) | ((uint32_t) (q[1] & 0xFF) << 8) | ((uint32_t) (q[2] & 0xFF) << 16) | ((uint32_t) (q[3] & 0xFF) << 24);
    const int ch = e >> 6, within = e & 63, plane = within >> 4, w = (within & 15) >> 2;
    pl[(((size_t) plane*T + t)*nch + ch)*4 + w] = packed;
    if ((e & 31) == 0)  bs[t*(nsb*8) + (e >> 5)] = s;
    if ((e & 255) == 0) dx[t*nsb + (e >> 8)] = d;
}

template <int WT, int T>
__device__ __forceinline__ void gemv_prologue(const GemvArgs & a, char * smem, int tid) {
    const int K = a.K, nb = K >> 5;
    const int wave = tid >> 6, lane = tid & 63;
    float * red = (float *) smem;                           // 64 floats scratch for reductions (first 256 B)
    char * base = smem + 256;
    uint32_t * lo = nullptr; uint32_t * hi = nullptr; float * dx = nullptr; int * sx = nullptr; uint16_t * xh = nullptr;
    if constexpr (WT == MI355X_TYPE_F16) {
        xh = (uint16_t *) base;
    } else if constexpr (WT == MI355X_TYPE_Q4_K) {
        lo = (uint32_t *) base;                             // planes
        dx = (float *) (base + (size_t) T*(K/64)*64);
        sx = (int *) (dx + T*(K/256));
    } else {
        lo = (uint32_t *) base;
        hi = lo + (size_t) T*nb*4;
        dx = (float *) (hi + (size_t) T*nb*4);
        sx = (int *) (dx + T*nb);
    }
    auto store4 = [&](const float v[4], int e, int t) {
        if constexpr (WT == MI355X_TYPE_F16) {
            *(uint2 *) (xh + (size_t) t*K + e) = make_uint2(f2h(v[0]) | ((uint32_t) f2h(v[1]) << 16), f2h(v[2]) | ((uint32_t) f2h(v[3]) << 16));
        } else if constexpr (WT == MI355X_TYPE_Q4_K) {
            q8_K_store(v, e, t, K, T, lo, dx, sx);
        } else {
            q8_0_store(v, e, t, nb, lo, hi, dx, sx);
        }
    };

    if (a.has_norm) {
        // K <= 2048: every thread owns float4 #tid and #tid+256 of each column
        float4 xv[T][2];
        float part[T];
        #pragma unroll
        for (int t = 0; t < T; t++) {
            const float * xr = (const float *) ((const char *) a.x + (int64_t) t*a.x_nb1);
            part[t] = 0.0f;
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                const int e = (j*256 + tid)*4;
                xv[t][j] = e < K ? *(const float4 *) (xr + e) : make_float4(0, 0, 0, 0);
                part[t] += (xv[t][j].x + xv[t][j].y) + (xv[t][j].z + xv[t][j].w);
            }
        }
        block_reduce_T(part, T, red, wave, lane);
        float mean[T];
        #pragma unroll
        for (int t = 0; t < T; t++) {
            mean[t] = part[t] / K;
            float v = 0.0f;
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                const int e = (j*256 + tid)*4;
                if (e < K) {
                    const float d0 = xv[t][j].x - mean[t], d1 = xv[t][j].y - mean[t], d2 = xv[t][j].z - mean[t], d3 = xv[t][j].w - mean[t];
                    v += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
                }
            }
            part[t] = v;
        }
        __syncthreads();
        block_reduce_T(part, T, red + 32, wave, lane);
        #pragma unroll
        for (int t = 0; t < T; t++) {
            const float sc = 1.0f / sqrtf(part[t] / K + a.eps);
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                const int e = (j*256 + tid)*4;
                if (e < K) {
                    const float4 w = *(const float4 *) (a.ln_w + e);
                    const float4 b = *(const float4 *) (a.ln_b + e);
                    float v[4];
                    v[0] = (xv[t][j].x - mean[t]) * sc; v[1] = (xv[t][j].y - mean[t]) * sc;
                    v[2] = (xv[t][j].z - mean[t]) * sc; v[3] = (xv[t][j].w - mean[t]) * sc;
                    v[0] = v[0]*w.x; v[1] = v[1]*w.y; v[2] = v[2]*w.z; v[3] = v[3]*w.w;
                    v[0] = v[0]+b.x; v[1] = v[1]+b.y; v[2] = v[2]+b.z; v[3] = v[3]+b.w;
                    store4(v, e, t);
                }
            }
        }
    } else {
        for (int t = 0; t < T; t++) {
            const float * xr = (const float *) ((const char *) a.x + (int64_t) t*a.x_nb1);
            for (int e = tid*4; e < K; e += 1024) {
                const float4 x4 = *(const float4 *) (xr + e);
                const float v[4] = { x4.x, x4.y, x4.z, x4.w };
                store4(v, e, t);
            }
        }
    }
    __syncthreads();
}

template <int WT, int T>
__global__ void __launch_bounds__(256) k_gemv(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = a.K, nb = K >> 5;

    gemv_prologue<WT, T>(a, smem, tid);

    char * base = smem + 256;
    const int ntot = a.row_start[a.nseg];
    const int row0 = (blockIdx.x * 4 + wave) * a.rows_per_wave;

    for (int rr = 0; rr < a.rows_per_wave; rr++) {
        const int grow = row0 + rr;
        if (grow >= ntot) break;
        int s = 0;
        if (a.nseg > 1 && grow >= a.row_start[1]) s = 1;
        if (a.nseg > 2 && grow >= a.row_start[2]) s = 2;
        const GemvSeg & sg = a.seg[s];
        const int row = grow - a.row_start[s];

        float acc[T];
        #pragma unroll
        for (int t = 0; t < T; t++) acc[t] = 0.0f;

        if constexpr (WT == MI355X_TYPE_F16) {
            const uint16_t * xh = (const uint16_t *) base;
            const uint4 * wrow = (const uint4 *) ((const char *) sg.w + (int64_t) row * K * 2);
            for (int g = lane; g < (K >> 3); g += 64) {
                const uint4 wq = wrow[g];
                const uint32_t ww[4] = { wq.x, wq.y, wq.z, wq.w };
                float wf[8];
                #pragma unroll
                for (int i = 0; i < 4; i++) { wf[2*i] = h2f((uint16_t) (ww[i] & 0xFFFF)); wf[2*i+1] = h2f((uint16_t) (ww[i] >> 16)); }
                #pragma unroll
                for (int t = 0; t < T; t++) {
                    const uint4 xq = *(const uint4 *) (xh + (size_t) t*K + g*8);
                    const uint32_t xw[4] = { xq.x, xq.y, xq.z, xq.w };
                    #pragma unroll
                    for (int i = 0; i < 4; i++) {
                        acc[t] = fmaf(wf[2*i],   h2f((uint16_t) (xw[i] & 0xFFFF)), acc[t]);
                        acc[t] = fmaf(wf[2*i+1], h2f((uint16_t) (xw[i] >> 16)),    acc[t]);
                    }
                }
            }
        } else if constexpr (WT == MI355X_TYPE_Q4_K) {
            const int nch = K >> 6, nsb = K >> 8;
            const uint4 * pl = (const uint4 *) base;
            const float * dx = (const float *) (base + (size_t) T*nch*64);
            const int *   bs = (const int *) (dx + T*nsb);
            const qplanes<MI355X_TYPE_Q4_K> p(sg.w, sg.nbt);
            float accm[T];
            #pragma unroll
            for (int t = 0; t < T; t++) accm[t] = 0.0f;
            for (int ch = lane; ch < nch; ch += 64) {
                const int sb = ch >> 2, c = ch & 3;
                const int64_t isb = (int64_t) row * nsb + sb;
                const uint4 q0 = *(const uint4 *) (p.qs + isb*128 + c*32);
                const uint4 q1 = *(const uint4 *) (p.qs + isb*128 + c*32 + 16);
                const uint32_t dm = p.dm[isb];
                const float dw = h2f((uint16_t) (dm & 0xFFFF)), dminw = h2f((uint16_t) (dm >> 16));
                int sc_lo, m_lo, sc_hi, m_hi;
                q4k_scale_min(2*c,     p.sc + isb*12, sc_lo, m_lo);
                q4k_scale_min(2*c + 1, p.sc + isb*12, sc_hi, m_hi);
                const uint32_t w[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
                #pragma unroll
                for (int t = 0; t < T; t++) {
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
                    acc[t]  = fmaf(dxv*dw, (float) isum, acc[t]);
                    accm[t] = fmaf(-dxv*dminw, (float) msum, accm[t]);
                }
            }
            #pragma unroll
            for (int t = 0; t < T; t++) acc[t] += accm[t];
        } else {
            const uint4 * lo = (const uint4 *) base;
            const uint4 * hi = lo + (size_t) T*nb;
            const float * dx = (const float *) (hi + (size_t) T*nb);
            const int *   sx = (const int *) (dx + T*nb);
            const qplanes<WT> p(sg.w, sg.nbt);
            for (int g = lane; g < nb; g += 64) {
                const int64_t ib = (int64_t) row * nb + g;
                uint32_t vlo[4], vhi[4];
                int off;
                if constexpr (WT == MI355X_TYPE_Q8_0) {
                    const uint4 q0 = *(const uint4 *) (p.qs + ib*32), q1 = *(const uint4 *) (p.qs + ib*32 + 16);
                    vlo[0] = q0.x; vlo[1] = q0.y; vlo[2] = q0.z; vlo[3] = q0.w;
                    vhi[0] = q1.x; vhi[1] = q1.y; vhi[2] = q1.z; vhi[3] = q1.w;
                    off = 0;
                } else {
                    const uint4 q = *(const uint4 *) (p.qs + ib*16);
                    const uint32_t w[4] = { q.x, q.y, q.z, q.w };
                    if constexpr (WT == MI355X_TYPE_Q5_0) {
                        const uint32_t qh = p.qh[ib];
                        #pragma unroll
                        for (int i = 0; i < 4; i++) {
                            vlo[i] = (w[i] & 0x0F0F0F0Fu)        | spread4_to_bit4(qh >> (4*i));
                            vhi[i] = ((w[i] >> 4) & 0x0F0F0F0Fu) | spread4_to_bit4(qh >> (16 + 4*i));
                        }
                        off = 16;
                    } else {
                        #pragma unroll
                        for (int i = 0; i < 4; i++) { vlo[i] = w[i] & 0x0F0F0F0Fu; vhi[i] = (w[i] >> 4) & 0x0F0F0F0Fu; }
                        off = 8;
                    }
                }
                const float dw = h2f(p.d[ib]);
                #pragma unroll
                for (int t = 0; t < T; t++) {
                    const uint4 al = lo[(size_t) t*nb + g], ah = hi[(size_t) t*nb + g];
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

        #pragma unroll
        for (int t = 0; t < T; t++) acc[t] = wave_sum(acc[t]);

        // epilogue: lane t finishes column t
        float v = acc[0];
        #pragma unroll
        for (int t = 1; t < T; t++) v = (lane == t) ? acc[t] : v;
        if (lane < T) {
            if (sg.bias)      v = v + sg.bias[row];
            if (sg.has_scale) v = v * sg.scale;
            if (sg.gelu)      v = gelu_lut(v, a.gelu_tab);
            if (sg.residual)  v = v + *(const float *) ((const char *) sg.residual + (int64_t) lane*sg.res_nb1 + (int64_t) row*4);
            char * dp = (char *) sg.dst + (int64_t) lane*sg.dst_nb1;
            if (sg.dst_f16) ((uint16_t *) dp)[row] = f2h(v); else ((float *) dp)[row] = v;
        }
    }
}

template <int WT>
static int launch_gemv_T(mi355x_ctx * ctx, const GemvArgs & k, int T, dim3 grid, uint32_t lds, double bytes, double flops) {
    const char * name = "gemv";
    switch (T) {
        case 1: return emit(ctx, name, k_gemv<WT, 1>, grid, dim3(256), lds, k, bytes, flops);
        case 2: return emit(ctx, name, k_gemv<WT, 2>, grid, dim3(256), lds, k, bytes, flops);
        case 3: return emit(ctx, name, k_gemv<WT, 3>, grid, dim3(256), lds, k, bytes, flops);
        case 4: return emit(ctx, name, k_gemv<WT, 4>, grid, dim3(256), lds, k, bytes, flops);
        case 5: return emit(ctx, name, k_gemv<WT, 5>, grid, dim3(256), lds, k, bytes, flops);
        case 6: return emit(ctx, name, k_gemv<WT, 6>, grid, dim3(256), lds, k, bytes, flops);
        case 7: return emit(ctx, name, k_gemv<WT, 7>, grid, dim3(256), lds, k, bytes, flops);
        case 8: return emit(ctx, name, k_gemv<WT, 8>, grid, dim3(256), lds, k, bytes, flops);
        default: return MI355X_E_UNSUPPORTED;
    }
}

extern "C" int mi355x_gemv_fused(mi355x_ctx * ctx, const mi355x_gemv_desc * d) {
    ctx->last_mirrored = 0;
    if (d->nseg < 1 || d->nseg > 3 || d->T < 1 || d->T > MI355X_MAX_COLS) return MI355X_E_UNSUPPORTED;
    // more than 8 columns: the plane kernels only (images of 8 columns, decode_q.hip)
    if (d->T > MI355X_IMG_COLS && !d->x_planes) return MI355X_E_UNSUPPORTED;
    for (int s = 0; s < d->nseg; s++) if (d->seg[s].ep.bias_per_col) return MI355X_E_UNSUPPORTED;      // (MFMA path only)
    if (d->x_planes) {   // wide cross-state batches: the matrix-core form (decode_mx.hip), same summation trees as the kernels below
        const int rc = mi355x_gemv_mx(ctx, d);
        if (rc != MI355X_E_UNSUPPORTED) return rc;
    }
    {   // the vocabulary projection has its own kernel (LayerNorm form or prepared planes)
        const int rc = mi355x_vocab(ctx, d);
        if (rc != MI355X_E_UNSUPPORTED) return rc;
    }
    if (d->x_planes) {   // activations already quantized (decode_q.hip pipeline): lean plane kernel, else the generic k_gemv8 on the same planes
        const int rc = mi355x_gemv_q(ctx, d);
        if (rc != MI355X_E_UNSUPPORTED) return rc;
        if (d->T <= MI355X_IMG_COLS) return mi355x_gemv8(ctx, d);
        // more than one image and no kernel that walks them itself (e.g. the vocabulary projection of a Q4_K model): image by image through
        // the <= 8-column paths — same per-column arithmetic, the weights are read once per image instead of once
        const int wt = d->seg[0].wtype;
        const size_t istride = mi355x_act_planes_bytes(wt, d->K, MI355X_IMG_COLS);          // (= one full image)
        bool mirrored = true;
        for (int c0 = 0; c0 < d->T; c0 += MI355X_IMG_COLS) {
            mi355x_gemv_desc sub = *d;
            mi355x_gemv_cols cols;
            sub.T = d->T - c0 < MI355X_IMG_COLS ? d->T - c0 : MI355X_IMG_COLS;
            sub.x_planes = (const char *) d->x_planes + (size_t) (c0 / MI355X_IMG_COLS) * istride;
            if (d->cols) {
                memset(&cols, 0, sizeof(cols));
                for (int s = 0; s < d->nseg; s++) for (int t = 0; t < sub.T; t++) { cols.dst[s][t] = d->cols->dst[s][c0 + t]; cols.res[s][t] = d->cols->res[s][c0 + t]; }
                for (int t = 0; t < sub.T; t++) { cols.mirror[t] = d->cols->mirror[c0 + t]; }
                sub.cols = &cols;
            } else {
                for (int s = 0; s < d->nseg; s++) {
                    if (sub.seg[s].dst) sub.seg[s].dst = (char *) sub.seg[s].dst + (int64_t) c0 * sub.seg[s].dst_nb1;
                    if (sub.seg[s].ep.residual) sub.seg[s].ep.residual = (const float *) ((const char *) sub.seg[s].ep.residual + (int64_t) c0 * sub.seg[s].ep.residual_nb1);
                }
            }
            if (d->planes_out) {
                int ntot = 0; for (int s = 0; s < d->nseg; s++) ntot += d->seg[s].N;
                sub.planes_out = (char *) d->planes_out + (size_t) (c0 / MI355X_IMG_COLS) * mi355x_act_planes_bytes(MI355X_TYPE_Q8_0, ntot, MI355X_IMG_COLS);
            }
            const int r2 = mi355x_gemv_fused(ctx, &sub);
            if (r2) return r2;              // (MI355X_E_UNSUPPORTED on the first image: nothing was launched; the <= 8-column paths do not differ between images)
            mirrored = mirrored && ctx->last_mirrored;
        }
        ctx->last_mirrored = mirrored ? 1 : 0;
        return 0;
    }
    {   // the lean decode kernels (decode.hip) take every quantized shape of the whisper graphs; what is left for k_gemv below:
        // F16 weights (f16 models), LDS-heavy shapes (K*T too large for the 64 KB planes)
        const int rc = mi355x_gemv8(ctx, d);
        if (rc != MI355X_E_UNSUPPORTED) return rc;
    }
    if (d->attn_part_o || !d->x) return MI355X_E_UNSUPPORTED;      // only the v2 kernel reads attention partials
    const int wt = d->seg[0].wtype, K = d->K, T = d->T;
    const int blk = wt == MI355X_TYPE_Q4_K ? 256 : (wt == MI355X_TYPE_F16 ? 8 : 32);
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0 && wt != MI355X_TYPE_Q4_K && wt != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
    if (K <= 0 || K % blk || K % 4 || ((uintptr_t) d->x % 16) || (d->x_nb1 % 16)) return MI355X_E_UNSUPPORTED;
    if (d->has_norm && (K > 2048 || !d->ln_w || !d->ln_b || ((uintptr_t) d->ln_w % 16) || ((uintptr_t) d->ln_b % 16))) return MI355X_E_UNSUPPORTED;
    const size_t lds = gemv_lds_bytes(wt, K, T);
    if (lds > 160*1024) return MI355X_E_UNSUPPORTED;

    GemvArgs k; memset(&k, 0, sizeof(k));
    k.x = d->x; k.x_nb1 = d->x_nb1; k.K = K; k.has_norm = d->has_norm; k.eps = d->eps; k.nseg = d->nseg;
    k.ln_w = d->ln_w; k.ln_b = d->ln_b; k.gelu_tab = ctx->gelu_tab;
    int ntot = 0; double wbytes = 0;
    for (int s = 0; s < d->nseg; s++) {
        const mi355x_gemv_seg & g = d->seg[s];
        if (g.wtype != wt || g.N <= 0 || ((uintptr_t) g.w % 16)) return MI355X_E_UNSUPPORTED;
        if (g.dst_type != MI355X_TYPE_F32 && g.dst_type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
        k.row_start[s] = ntot;
        GemvSeg & o = k.seg[s];
        o.w = g.w; o.N = g.N; o.nbt = wt == MI355X_TYPE_F16 ? 0 : (int64_t) g.N * (K / blk);
        o.bias = g.ep.bias; o.scale = g.ep.scale; o.has_scale = g.ep.has_scale; o.gelu = g.ep.gelu;
        o.residual = g.ep.residual; o.res_nb1 = g.ep.residual_nb1;
        o.dst = g.dst; o.dst_nb1 = g.dst_nb1; o.dst_f16 = g.dst_type == MI355X_TYPE_F16;
        ntot += g.N;
        wbytes += wt == MI355X_TYPE_F16 ? (double) g.N * K * 2 : (double) mi355x_type_row_bytes(wt, K) * g.N;
    }
    for (int s = d->nseg; s < 4; s++) k.row_start[s] = ntot;
    // spread rows over ~8 waves per CU; every row is streamed exactly once
    int rpw = (ntot + ctx->n_cu*8 - 1) / (ctx->n_cu*8);
    if (rpw < 1) rpw = 1; if (rpw > 16) rpw = 16;
    k.rows_per_wave = rpw;
    const int nblocks = (ntot + 4*rpw - 1) / (4*rpw);
    const double bytes = wbytes + (double) K*T*4 + (double) ntot*T*4;
    const double flops = 2.0 * ntot * K * T;
    switch (wt) {
        case MI355X_TYPE_Q4_0: return launch_gemv_T<MI355X_TYPE_Q4_0>(ctx, k, T, dim3(nblocks), (uint32_t) lds, bytes, flops);
        case MI355X_TYPE_Q5_0: return launch_gemv_T<MI355X_TYPE_Q5_0>(ctx, k, T, dim3(nblocks), (uint32_t) lds, bytes, flops);
        case MI355X_TYPE_Q8_0: return launch_gemv_T<MI355X_TYPE_Q8_0>(ctx, k, T, dim3(nblocks), (uint32_t) lds, bytes, flops);
        case MI355X_TYPE_Q4_K: return launch_gemv_T<MI355X_TYPE_Q4_K>(ctx, k, T, dim3(nblocks), (uint32_t) lds, bytes, flops);
        case MI355X_TYPE_F16:  return launch_gemv_T<MI355X_TYPE_F16>(ctx, k, T, dim3(nblocks), (uint32_t) lds, bytes, flops);
    }
    return MI355X_E_UNSUPPORTED;
}
