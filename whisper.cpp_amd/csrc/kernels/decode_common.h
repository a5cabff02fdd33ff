// Shared pieces of the decoder-step kernels (decode.hip: the fused T <= 2 family and the generic k_gemv8; decode_q.hip: the
// pre-quantized-activation pipeline for T >= 3 columns and for cross-state batches): kernel argument blocks, planar weight-block
// loads, the Q8_0 / Q8_K activation quantizers and the LDS / global layout of the quantized activation planes.
#pragma once
#include "common.h"
#include <math.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct DGSeg {
    const void *  w;  int64_t nbt;  int N;  int has_scale;
    const float * bias; float scale; int gelu;
    const float * residual; int64_t res_nb1;
    void * dst; int64_t dst_nb1; int dst_f16; int pad;
};
struct DGArgs {
    const float * x; int64_t x_nb1; int K; int has_norm; float eps; int nseg;
    const float * ln_w; const float * ln_b;
    const float * part_o; const float * part_ml; int nparts; int passes;     // x == nullptr: x = combine(attention partials)
    int row_start[4];
    int xfirst; int ntot;
    DGSeg seg[3];
    const uint16_t * gelu_tab;
    unsigned long long * dbg;          // GGML_MI355X_KTIME=1: s_memtime stamps of workgroup 0 / wave 0 (kernel anatomy, scripts/kbench.py)
    // k_gemv8 only (appended: the offsets of everything above are what the tuned k_gemv_row family was built with):
    const void * xq;                   // activations already quantized: image of the LDS planes for (K, T) (decode_q.hip), copied instead of computed
    void * dstcol[8];                  // use_cols: column t of segment 0 is stored at dstcol[t] instead of dst + t*dst_nb1 (cross-state batches)
    int use_cols; int pad2;
    void * mircol[8];                  // mircol[0] != NULL: column t of segment 0 is also stored (F32) at mircol[t] (host-visible mirror of the logits)
};
// Kernel-anatomy stamps are compiled in only with -DMI355X_KTIME (MI355X_KTIME_BUILD=1 python whisper.cpp_amd/build.py): even
// with a null pointer each of the seven stamp sites costs a saveexec / branch / restore triple and a basic-block boundary in
// kernels whose whole body is ~2 us.
#ifdef MI355X_KTIME
#define DG_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DG_STAMP(i) do { } while (0)
#endif

// LPR = lanes per weight row (8/16/32/64 => 8/4/2/1 rows per wave pass).  Decode mat-vecs are latency-bound: what counts
// is how many waves have loads in flight right after launch, so small matrices use more lanes (= more waves) per row and
// only the big ones (logits) use the 8-lane layout.  DG_U(LPR) = 32-element blocks per lane per chunk: LPR*DG_U blocks
// cover a K = 1280 row in one chunk for every LPR and a K = 5120 row in one (LPR 64) to four (LPR 8) chunks.
#define DG_U(LPR) ((LPR) == 8 ? 5 : 3)
#define DG_XR 5           // float4 activation registers per thread in the "activations first" order

static inline size_t dg_lds_bytes(int wt, int K, int T, bool staged) {
    // red[64 floats] | lo[T][nb] uint4 | hi[T][nb] uint4 | dx[T][nb] f32 | sx[T][nb] i32 | stage[T][K] f32 (optional)
    // Q4_K: red | four Q8_K planes [4][T][K/64] uint4 | dx[T][K/256] f32 | sums[T][K/32] i32 | (16-byte aligned) stage
    const size_t act = wt == MI355X_TYPE_Q4_K ? (((size_t) T * ((size_t) K + (K/256)*4 + (K/32)*4) + 15) & ~(size_t) 15) : (size_t) T * (K/32) * 40;
    return 256 + act + (staged ? (size_t) T * K * 4 : 0);
}

template <int WT> struct wblk;
template <> struct wblk<MI355X_TYPE_Q4_0> { u32x4 q; uint16_t d; };
template <> struct wblk<MI355X_TYPE_Q5_0> { u32x4 q; uint32_t qh; uint16_t d; };
template <> struct wblk<MI355X_TYPE_Q8_0> { u32x4 q, q1; uint16_t d; };

template <int WT>
__device__ __forceinline__ void wblk_load(wblk<WT> & r, const char * base, int64_t nbt, int64_t ib, bool ok) {
    const u32x4 z = { 0, 0, 0, 0 };
    if constexpr (WT == MI355X_TYPE_Q8_0) {
        const u32x4 * q = (const u32x4 *) (base + ib*32);
        r.q  = ok ? __builtin_nontemporal_load(q)     : z;
        r.q1 = ok ? __builtin_nontemporal_load(q + 1) : z;
        r.d  = ok ? *((const uint16_t *) (base + nbt*32) + ib) : (uint16_t) 0;    // small planes: plain loads (L1 reuse across chunks)
    } else {
        r.q = ok ? __builtin_nontemporal_load((const u32x4 *) (base + ib*16)) : z;
        if constexpr (WT == MI355X_TYPE_Q5_0) {
            r.qh = ok ? *((const uint32_t *) (base + nbt*16) + ib) : 0u;
            r.d  = ok ? *((const uint16_t *) (base + nbt*20) + ib) : (uint16_t) 0;
        } else {
            r.d  = ok ? *((const uint16_t *) (base + nbt*16) + ib) : (uint16_t) 0;
        }
    }
}

// unconditional form: the caller passes an always-valid block index (clamped) and zeroes the scale of duplicates
template <int WT>
__device__ __forceinline__ void wblk_load(wblk<WT> & r, const char * base, int64_t nbt, int64_t ib) {
    if constexpr (WT == MI355X_TYPE_Q8_0) {
        const u32x4 * q = (const u32x4 *) (base + ib*32);
        r.q  = __builtin_nontemporal_load(q);
        r.q1 = __builtin_nontemporal_load(q + 1);
        r.d  = *((const uint16_t *) (base + nbt*32) + ib);
    } else {
        r.q = __builtin_nontemporal_load((const u32x4 *) (base + ib*16));
        if constexpr (WT == MI355X_TYPE_Q5_0) {
            r.qh = *((const uint32_t *) (base + nbt*16) + ib);
            r.d  = *((const uint16_t *) (base + nbt*20) + ib);
        } else {
            r.d  = *((const uint16_t *) (base + nbt*16) + ib);
        }
    }
}

// integer dot of one weight block with the Q8_0 activation block (al = elements 0..15, ah = 16..31), minus the offset term
template <int WT>
__device__ __forceinline__ void wblk_unpack(const wblk<WT> & r, uint32_t vlo[4], uint32_t vhi[4]) {
    if constexpr (WT == MI355X_TYPE_Q8_0) {
        #pragma unroll
        for (int i = 0; i < 4; i++) { vlo[i] = r.q[i]; vhi[i] = r.q1[i]; }
    } else if constexpr (WT == MI355X_TYPE_Q5_0) {
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            vlo[i] = (r.q[i] & 0x0F0F0F0Fu)        | spread4_to_bit4(r.qh >> (4*i));
            vhi[i] = ((r.q[i] >> 4) & 0x0F0F0F0Fu) | spread4_to_bit4(r.qh >> (16 + 4*i));
        }
    } else {
        #pragma unroll
        for (int i = 0; i < 4; i++) { vlo[i] = r.q[i] & 0x0F0F0F0Fu; vhi[i] = (r.q[i] >> 4) & 0x0F0F0F0Fu; }
    }
}

// quantize 4 consecutive values (one lane of an 8-lane group = one 32-block) to Q8_0 and store to LDS
__device__ __forceinline__ void dg_q8_0_store(const float v[4], int e, int t, int nb, uint32_t * lo, uint32_t * hi, float * dx, int * sx) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = group_max<8>(amax);
    const float d  = amax / 127.0f;
    const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
    const int q0 = (int) rintf(v[0]*id), q1 = (int) rintf(v[1]*id), q2 = (int) rintf(v[2]*id), q3 = (int) rintf(v[3]*id);
    int s = group_sum_i<8>(q0 + q1 + q2 + q3);
    const uint32_t packed = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
    const int b = e >> 5, w = (e & 31) >> 2;
    uint32_t * plane = w < 4 ? lo : hi;
    plane[((size_t) t*nb + b)*4 + (w & 3)] = packed;
    if (w == 0) { dx[t*nb + b] = round_f16(d); sx[t*nb + b] = s; }
}

// ---- Q4_K support of the lean decode kernels -------------------------------------------------------------------
// weights: one lane-unit = 64 elements = 32 bytes of nibbles (sub-blocks 2c: low nibbles, 2c+1: high nibbles) of a
// 256-element super-block (ggml-common.h:327-338); activations: Q8_K (ggml-quants.c:2768-2805) in four 16-byte planes per
// 64-element chunk + per-super-block scale + per-32-element sums (for the mins), as in the first-generation k_gemv.
template <> struct wblk<MI355X_TYPE_Q4_K> { u32x4 q, q1; uint32_t dm; uint32_t sc[3]; uint16_t d; };

// unit index ch (64-element chunk) of row `row`; nbt = total super-blocks of the tensor, nsb = super-blocks per row
__device__ __forceinline__ void wblk_load_q4k(wblk<MI355X_TYPE_Q4_K> & r, const char * base, int64_t nbt, int64_t row, int nsb, int ch) {
    const int64_t isb = row * nsb + (ch >> 2);
    const u32x4 * q = (const u32x4 *) (base + isb*128 + (ch & 3)*32);
    r.q  = __builtin_nontemporal_load(q);
    r.q1 = __builtin_nontemporal_load(q + 1);
    const uint32_t * sc = (const uint32_t *) (base + nbt*128 + isb*12);
    r.sc[0] = sc[0]; r.sc[1] = sc[1]; r.sc[2] = sc[2];
    r.dm = *((const uint32_t *) (base + nbt*140) + isb);
    r.d = 0;
}

// quantize 4 consecutive values (one lane of a WAVE = one 256-element super-block) to Q8_K and store to LDS
__device__ __forceinline__ void dg_q8_K_store(const float v[4], int e, int t, int K, int T, uint32_t * pl, float * dx, int * bs) {
    const float mx = group_max<64>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    const float mn = -group_max<64>(-fminf(fminf(v[0], v[1]), fminf(v[2], v[3])));
    const float amax = fmaxf(mx, -mn);
    const float maxv = (mx >= -mn) ? mx : mn;                          // value with the largest magnitude (sign kept)
    const int nch = K >> 6, nsb = K >> 8;
    int q[4] = { 0, 0, 0, 0 };
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.0f / maxv;
        #pragma unroll
        for (int i = 0; i < 4; i++) { const int r = (int) rintf(iscale * v[i]); q[i] = r < 127 ? r : 127; }
        d = 1.0f / iscale;
    }
    const int s = group_sum_i<8>(q[0] + q[1] + q[2] + q[3]);             // sum over 32 elements
    const uint32_t packed = (uint32_t) (q[0] & 0xFF) | ((uint32_t) (q[1] & 0xFF) << 8) | ((uint32_t) (q[2] & 0xFF) << 16) | ((uint32_t) (q[3] & 0xFF) << 24);
    const int ch = e >> 6, within = e & 63, plane = within >> 4, w = (within & 15) >> 2;
    pl[(((size_t) plane*T + t)*nch + ch)*4 + w] = packed;
    if ((e & 31) == 0)  bs[t*(nsb*8) + (e >> 5)] = s;
    if ((e & 255) == 0) dx[t*nsb + (e >> 8)] = d;
}

// dot of one 64-element weight unit with the Q8_K activations of column t: acc += dx*d*isum, accm += -dx*dmin*msum
// (vec_dot q4_K x q8_K: ggml-cpu/quants.c:696-769)
template <int T>
__device__ __forceinline__ void wblk_dot_q4k(const wblk<MI355X_TYPE_Q4_K> & r, int ch, float live, int nch, int nsb,
                                             const uint4 * pl, const float * dx, const int * bs, float * acc, float * accm) {
    const int sb = ch >> 2, c = ch & 3;
    const float dw = h2f((uint16_t) (r.dm & 0xFFFF)) * live, dminw = h2f((uint16_t) (r.dm >> 16)) * live;
    int sc_lo, m_lo, sc_hi, m_hi;
    q4k_scale_min_w(2*c,     r.sc[0], r.sc[1], r.sc[2], sc_lo, m_lo);
    q4k_scale_min_w(2*c + 1, r.sc[0], r.sc[1], r.sc[2], sc_hi, m_hi);
    const uint32_t w[8] = { r.q[0], r.q[1], r.q[2], r.q[3], r.q1[0], r.q1[1], r.q1[2], r.q1[3] };
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

// LDS bytes of the activation planes: Q8_0 family 40 B per 32 elements, Q8_K K + 4*K/256 + 4*K/32 bytes per column
static inline __host__ __device__ size_t dg_act_bytes(int wt, int K, int T) {
    return wt == MI355X_TYPE_Q4_K ? (size_t) T * ((size_t) K + (K/256)*4 + (K/32)*4) : (size_t) T * (K/32) * 40;
}

// Planes of more than 8 columns (cross-state batches): ceil(T/8) images of <= 8 columns back to back, image g at g * dg_img_stride
// (mi355x_kernels.h: MI355X_IMG_COLS).  T <= 8: one image, the layout above.
static inline __host__ __device__ size_t dg_img_stride(int wt, int K) { return (dg_act_bytes(wt, K, MI355X_IMG_COLS) + 15) & ~(size_t) 15; }
static inline __host__ __device__ size_t dg_planes_bytes(int wt, int K, int T) {
    const int G = (T + MI355X_IMG_COLS - 1) / MI355X_IMG_COLS;
    return (size_t) (G - 1) * dg_img_stride(wt, K) + ((dg_act_bytes(wt, K, T - MI355X_IMG_COLS*(G - 1)) + 15) & ~(size_t) 15);
}

// waves (= rows) per workgroup of k_gemv_row: enough threads for one float4 activation slot each when K <= 2048
static inline int gemv_row_waves(int K) {
    if (K > 2048) return 4;
    const int w = (K/4 + 63) / 64;
    return w < 4 ? 4 : (w > 8 ? 8 : w);
}

