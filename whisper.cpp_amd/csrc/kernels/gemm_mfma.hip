// Dense half of the hot path (encoder, prompt, conv-as-GEMM): tiled GEMM on the CDNA4 matrix cores.
//
//   dst[m, t] = sum_k A[m, k] * B[t, k]       A = ggml src0 (weights, M rows), B = prepared activations
//
//  * A is block-quantized (planar Q4_0/Q5_0/Q8_0/Q4_K) or F16/F32.  Each K-step a workgroup dequantizes its
//    128 x 64 A tile in registers and stores it as f16 into an XOR-swizzled LDS tile ("per-warp dequant into LDS
//    tiles feeding MFMA"); the HBM side stays in the quantized layout (0.56-1.06 B/weight).
//  * B is an f16 [T][K] matrix produced by k_prep_act below.  For quantized A it holds the reference's OWN
//    activation rounding: x is quantized to Q8_0 / Q8_K blocks exactly as ggml-cpu does before its integer dot
//    (ggml-cpu/ggml-cpu.c:1322-1357; arch/x86/quants.c:302-398; ggml-quants.c:2768-2805) and stored as
//    f16(d * q).  So the MFMA path sees the same quantization decisions as the CPU path; what differs is f16
//    rounding of d*q products (2^-12 relative) and f32 summation order.
//  * v_mfma_f32_32x32x16_f16, f32 accumulate.  4 waves per workgroup in a 2x2 arrangement, each wave owns a
//    64 x (BN/2) output tile; A = weights so consecutive accumulator registers are consecutive output features
//    (contiguous in dst) => 16-byte stores.
//  * Global loads for K-step k+1 are issued before the MFMAs of step k (register-staged prefetch), LDS tiles
//    are conflict-free for ds_read_b128 via slot ^= (row>>1)&7.
#include "common.h"
#include "qrows.h"
#include <atomic>

// -------------------------------------------------------------------------------------------------
// activation preparation: f32 [K, T] (ggml src1) -> f16 [T][K]
//   mode 0: plain f16 rounding (F16/F32 weights: vec_dot_type F16, ggml-cpu/ggml-cpu.c:214-412)
//   mode 1: Q8_0 round trip     mode 2: Q8_K round trip
// -------------------------------------------------------------------------------------------------
//   mode 3 / 4: the Q8_0 / Q8_K blocks themselves (int8 + scales: the activation ROWS of the int8 tile GEMM, qrows.h / mmq.hip)
struct PrepArgs { const char * x; int64_t x_nb1; uint16_t * y; int K; int64_t T; int mode; int x_f16; int8_t * rq; float * rd; int * rs; };

__global__ void __launch_bounds__(256) k_prep_act(const PrepArgs a) {
    const int64_t t = blockIdx.y;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= a.K) return;
    float v[4];
    if (a.x_f16) {
        const uint2 h = *(const uint2 *) (a.x + t*a.x_nb1 + (int64_t) e*2);
        v[0] = h2f((uint16_t) (h.x & 0xFFFF)); v[1] = h2f((uint16_t) (h.x >> 16)); v[2] = h2f((uint16_t) (h.y & 0xFFFF)); v[3] = h2f((uint16_t) (h.y >> 16));
    } else {
        const float4 x4 = *(const float4 *) (a.x + t*a.x_nb1 + (int64_t) e*4);
        v[0] = x4.x; v[1] = x4.y; v[2] = x4.z; v[3] = x4.w;
    }
    float r[4];
    if (a.mode == MI355X_PREP_Q8_0_ROWS) {
        // quantize_row_q8_0 (arch/x86/quants.c:302-398): the statements of dg_q8_0_store (decode_common.h); 8 lanes = one block
        float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
        const float d  = amax / 127.0f;
        const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
        const int q0 = (int) rintf(v[0]*id), q1 = (int) rintf(v[1]*id), q2 = (int) rintf(v[2]*id), q3 = (int) rintf(v[3]*id);
        *(uint32_t *) (a.rq + t*a.K + e) = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
        if ((e & 31) == 0) a.rd[(int64_t) (e >> 5)*a.T + t] = round_f16(d);
        return;
    }
    if (a.mode == MI355X_PREP_Q8_K_ROWS) {
        // quantize_row_q8_K (ggml-quants.c:2768-2805): the statements of dg_q8_K_store; the wave = one 256-element super-block
        float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
        const float amax = fmaxf(mx, -mn);
        const float maxv = (mx >= -mn) ? mx : mn;
        int q[4] = { 0, 0, 0, 0 };
        float d = 0.0f;
        if (amax != 0.0f) {
            const float iscale = -127.0f / maxv;
            #pragma unroll
            for (int i = 0; i < 4; i++) { const int qi = (int) rintf(iscale * v[i]); q[i] = qi < 127 ? qi : 127; }
            d = 1.0f / iscale;
        }
        int s = q[0] + q[1] + q[2] + q[3];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        *(uint32_t *) (a.rq + t*a.K + e) = (uint32_t) (q[0] & 0xFF) | ((uint32_t) (q[1] & 0xFF) << 8) | ((uint32_t) (q[2] & 0xFF) << 16) | ((uint32_t) (q[3] & 0xFF) << 24);
        if ((e & 31) == 0)  a.rs[(int64_t) (e >> 5)*a.T + t] = s;
        if ((e & 255) == 0) a.rd[(int64_t) (e >> 8)*a.T + t] = d;
        return;
    }
    if (a.mode == 0) {
        #pragma unroll
        for (int i = 0; i < 4; i++) r[i] = v[i];
    } else if (a.mode == 1) {
        float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
        const float d  = round_f16(amax / 127.0f);
        const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
        #pragma unroll
        for (int i = 0; i < 4; i++) r[i] = d * rintf(v[i]*id);
    } else {
        float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
        const float amax = fmaxf(mx, -mn);
        const float maxv = (mx >= -mn) ? mx : mn;
        if (amax == 0.0f) { r[0] = r[1] = r[2] = r[3] = 0.0f; }
        else {
            const float iscale = -127.0f / maxv;
            const float d = 1.0f / iscale;
            #pragma unroll
            for (int i = 0; i < 4; i++) r[i] = d * fminf(127.0f, rintf(iscale * v[i]));
        }
    }
    if (a.mode != 0) {
        // d*q is stored as f16: the reference keeps d and q apart and stays finite up to ~8e6 where f16 overflows at 65504.
        // Saturate instead of producing inf (mode 0 keeps the reference's own f16 conversion of the activation, inf included).
        #pragma unroll
        for (int i = 0; i < 4; i++) r[i] = fminf(fmaxf(r[i], -65504.0f), 65504.0f);
    }
    *(uint2 *) (a.y + t*a.K + e) = make_uint2(f2h(r[0]) | ((uint32_t) f2h(r[1]) << 16), f2h(r[2]) | ((uint32_t) f2h(r[3]) << 16));
}

extern "C" int mi355x_prep_act(mi355x_ctx * ctx, const void * x, int64_t x_nb1, int x_f16, void * yv, int K, int64_t T, int mode) {
    uint16_t * y = (uint16_t *) yv;
    if (mode < 0 || mode > 4 || K % 4 || (mode == 1 && K % 32) || (mode == 2 && K % 256) || T > 65535*64LL) return MI355X_E_UNSUPPORTED;
    if (mode >= 3 && (K % 128 || (mode == 4 && K % 256) || ((uintptr_t) yv % 16) || T > 65535)) return MI355X_E_UNSUPPORTED;
    PrepArgs k = { (const char *) x, x_nb1, y, K, T, mode, x_f16, nullptr, nullptr, nullptr };
    if (mode >= 3) { const qrows_t R = qrows_of(yv, mode == 4, K, T); k.rq = R.q; k.rd = R.d; k.rs = R.bsum; }
    // blockIdx.y limited to 65535: loop in chunks
    int rc = 0;
    for (int64_t t0 = 0; t0 < T && rc == 0; t0 += 65535) {
        PrepArgs kk = k; kk.x += t0*x_nb1; kk.y += t0*K;
        const int64_t nt = T - t0 < 65535 ? T - t0 : 65535;
        rc = emit(ctx, "prep_act", k_prep_act, dim3((uint32_t) ((K/4 + 255) / 256), (uint32_t) nt), dim3(256), 0, kk, (double) nt*K*(x_f16 ? 4 : 6), 0);
    }
    return rc;
}

// -------------------------------------------------------------------------------------------------
// GEMM
// -------------------------------------------------------------------------------------------------
struct GemmArgs {
    const char * A; int64_t a_nb1; int64_t nbt;        // A rows: planar quant (nbt = total blocks) or f16/f32 with byte row stride
    const uint16_t * B; int64_t ldb;                    // f16 [T][ldb]
    int M, K; int64_t T;
    char * dst; int64_t dst_nb1; int dst_f16;
    const float * bias; float scale; int has_scale; int gelu; int bias_t;   // bias_t: bias is per column t (mi355x_epilogue::bias_per_col)
    const char * residual; int64_t res_nb1;
    const uint16_t * gelu_tab;
    int mt, nt, per, m_major;                           // k_gemm_f16_ring: tile counts and the XCD-aware tile order (launch_ring)
    uint16_t * prep; int prep_only;                     // epilogue also writes k_prep_act(mode 1) of the result, f16 [T][M]; prep_only: no dst store
    int ablate;                                         // k_gemm_dq, timing experiments only (MI355X_OPT_DQ_ABLATE): 1 no unpack, 2 no B DMA, 4 no MFMA, 8 no A loads
    int lds_bytes;                                      // k_gemm_dq: dynamic LDS of this launch (the GELU table moves into LDS when it fits behind the transposition regions)
    unsigned long long * dbg;                           // GGML_MI355X_KTIME=1: s_memtime stamps of workgroup 0, waves 0 and NW-1 (kernel anatomy, scripts/gemm_kbench.py)
};

#define BM 128
#define BK 64

// swizzled byte offset of 16-byte slot `slot` (0..7) of row `row` in a [rows][64] f16 tile
__device__ __forceinline__ int lds_off(int row, int slot) { return row*128 + ((slot ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint32_t pack_h2(float a, float b) { return (uint32_t) f2h(a) | ((uint32_t) f2h(b) << 16); }

// registers holding one thread's share of the next A tile
template <int AT> struct a_regs;
template <> struct a_regs<MI355X_TYPE_Q4_0> { uint4 q; uint16_t d; };
template <> struct a_regs<MI355X_TYPE_Q5_0> { uint4 q; uint32_t qh; uint16_t d; };
template <> struct a_regs<MI355X_TYPE_Q8_0> { uint4 q0, q1; uint16_t d; };
template <> struct a_regs<MI355X_TYPE_Q4_K> { uint4 q0, q1; uint32_t dm; uint32_t sc[3]; };
template <> struct a_regs<MI355X_TYPE_F16>  { uint4 v[4]; };
template <> struct a_regs<MI355X_TYPE_F32>  { float4 v[8]; };

// thread (row = tid>>1, half = tid&1) owns 32 consecutive k of its row: k0 + half*32 .. +31
template <int AT>
__device__ __forceinline__ void a_load(a_regs<AT> & r, const GemmArgs & a, int m, int k, bool valid) {
    if constexpr (AT == MI355X_TYPE_F16) {
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool ok = valid && (k + 8*i < a.K);
            r.v[i] = ok ? *(const uint4 *) (a.A + (int64_t) m*a.a_nb1 + (int64_t) (k + 8*i)*2) : make_uint4(0, 0, 0, 0);
        }
    } else if constexpr (AT == MI355X_TYPE_F32) {
        #pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool ok = valid && (k + 4*i < a.K);
            r.v[i] = ok ? *(const float4 *) (a.A + (int64_t) m*a.a_nb1 + (int64_t) (k + 4*i)*4) : make_float4(0, 0, 0, 0);
        }
    } else if constexpr (AT == MI355X_TYPE_Q4_K) {
        const qplanes<MI355X_TYPE_Q4_K> p(a.A, a.nbt);
        const bool ok = valid && (k < a.K);
        if (ok) {
            const int64_t sb = (int64_t) m * (a.K >> 8) + (k >> 8);
            const int j = (k & 255) >> 5;
            const uint8_t * qs = p.qs + sb*128 + (j >> 1)*32;
            r.q0 = *(const uint4 *) qs; r.q1 = *(const uint4 *) (qs + 16);
            r.dm = p.dm[sb];
            const uint32_t * sc = (const uint32_t *) (p.sc + sb*12);
            r.sc[0] = sc[0]; r.sc[1] = sc[1]; r.sc[2] = sc[2];
        } else { r.q0 = r.q1 = make_uint4(0, 0, 0, 0); r.dm = 0; r.sc[0] = r.sc[1] = r.sc[2] = 0; }
    } else {
        const qplanes<AT> p(a.A, a.nbt);
        const bool ok = valid && (k < a.K);
        const int64_t ib = (int64_t) m * (a.K >> 5) + (k >> 5);
        if constexpr (AT == MI355X_TYPE_Q8_0) {
            r.q0 = ok ? *(const uint4 *) (p.qs + ib*32) : make_uint4(0, 0, 0, 0);
            r.q1 = ok ? *(const uint4 *) (p.qs + ib*32 + 16) : make_uint4(0, 0, 0, 0);
        } else {
            r.q = ok ? *(const uint4 *) (p.qs + ib*16) : make_uint4(0, 0, 0, 0);
            if constexpr (AT == MI355X_TYPE_Q5_0) r.qh = ok ? p.qh[ib] : 0;
        }
        r.d = ok ? p.d[ib] : (uint16_t) 0;
    }
}

// dequantize this thread's 32 k-values to f16: 4 slots of 8 halves
template <int AT>
__device__ __forceinline__ void a_unpack(const a_regs<AT> & r, int k /* global k of first element */, uint4 (&out)[4]) {
    if constexpr (AT == MI355X_TYPE_F16) {
        #pragma unroll
        for (int i = 0; i < 4; i++) out[i] = r.v[i];
    } else if constexpr (AT == MI355X_TYPE_F32) {
        #pragma unroll
        for (int i = 0; i < 4; i++)
            out[i] = make_uint4(pack_h2(r.v[2*i].x, r.v[2*i].y), pack_h2(r.v[2*i].z, r.v[2*i].w), pack_h2(r.v[2*i+1].x, r.v[2*i+1].y), pack_h2(r.v[2*i+1].z, r.v[2*i+1].w));
    } else if constexpr (AT == MI355X_TYPE_Q8_0) {
        const float d = h2f(r.d);
        const uint32_t w[8] = { r.q0.x, r.q0.y, r.q0.z, r.q0.w, r.q1.x, r.q1.y, r.q1.z, r.q1.w };
        uint32_t o[16];
        #pragma unroll
        for (int i = 0; i < 8; i++) {
            o[2*i]   = pack_h2((float) (int8_t) (w[i] & 0xFF) * d,         (float) (int8_t) ((w[i] >> 8) & 0xFF) * d);
            o[2*i+1] = pack_h2((float) (int8_t) ((w[i] >> 16) & 0xFF) * d, (float) (int8_t) (w[i] >> 24) * d);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) out[i] = make_uint4(o[4*i], o[4*i+1], o[4*i+2], o[4*i+3]);
    } else if constexpr (AT == MI355X_TYPE_Q4_K) {
        const float d = h2f((uint16_t) (r.dm & 0xFFFF)), dmin = h2f((uint16_t) (r.dm >> 16));
        const int j = (k & 255) >> 5;
        int sc, m; q4k_scale_min_w(j, r.sc[0], r.sc[1], r.sc[2], sc, m);
        const float d1 = d * sc, m1 = dmin * m;
        const int sh = (j & 1) * 4;
        const uint32_t w[8] = { r.q0.x, r.q0.y, r.q0.z, r.q0.w, r.q1.x, r.q1.y, r.q1.z, r.q1.w };
        uint32_t o[16];
        #pragma unroll
        for (int i = 0; i < 8; i++) {
            const float v0 = d1 * (float) ((w[i] >> (sh))      & 0xF) - m1, v1 = d1 * (float) ((w[i] >> (8 + sh))  & 0xF) - m1;
            const float v2 = d1 * (float) ((w[i] >> (16 + sh)) & 0xF) - m1, v3 = d1 * (float) ((w[i] >> (24 + sh)) & 0xF) - m1;
            o[2*i] = pack_h2(v0, v1); o[2*i+1] = pack_h2(v2, v3);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) out[i] = make_uint4(o[4*i], o[4*i+1], o[4*i+2], o[4*i+3]);
    } else {
        // Q4_0 / Q5_0: byte j of qs -> element j (low nibble) and j+16 (high nibble)
        const float d = h2f(r.d);
        const uint32_t w[4] = { r.q.x, r.q.y, r.q.z, r.q.w };
        uint32_t lo[4], hi[4];
        int off;
        if constexpr (AT == MI355X_TYPE_Q5_0) {
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                lo[i] = (w[i] & 0x0F0F0F0Fu)        | spread4_to_bit4(r.qh >> (4*i));
                hi[i] = ((w[i] >> 4) & 0x0F0F0F0Fu) | spread4_to_bit4(r.qh >> (16 + 4*i));
            }
            off = 16;
        } else {
            #pragma unroll
            for (int i = 0; i < 4; i++) { lo[i] = w[i] & 0x0F0F0F0Fu; hi[i] = (w[i] >> 4) & 0x0F0F0F0Fu; }
            off = 8;
        }
        uint32_t o[16];
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            o[2*i]       = pack_h2((float) ((int) (lo[i] & 0xFF) - off) * d,         (float) ((int) ((lo[i] >> 8) & 0xFF) - off) * d);
            o[2*i+1]     = pack_h2((float) ((int) ((lo[i] >> 16) & 0xFF) - off) * d, (float) ((int) (lo[i] >> 24) - off) * d);
            o[8 + 2*i]   = pack_h2((float) ((int) (hi[i] & 0xFF) - off) * d,         (float) ((int) ((hi[i] >> 8) & 0xFF) - off) * d);
            o[8 + 2*i+1] = pack_h2((float) ((int) ((hi[i] >> 16) & 0xFF) - off) * d, (float) ((int) (hi[i] >> 24) - off) * d);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) out[i] = make_uint4(o[4*i], o[4*i+1], o[4*i+2], o[4*i+3]);
    }
}

template <int AT>
__device__ __forceinline__ void a_store(const a_regs<AT> & r, char * lds, int row, int half, int k) {
    uint4 out[4];
    a_unpack<AT>(r, k, out);
    #pragma unroll
    for (int i = 0; i < 4; i++) *(uint4 *) (lds + lds_off(row, half*4 + i)) = out[i];
}

// -------------------------------------------------------------------------------------------------
// one-time weight preparation for the MFMA path: planar quantized A -> f16 [M][K] (the values k_gemm_mfma would
// put into LDS, bit for bit).  With 288 GB of HBM the backend keeps this copy next to the quantized one for
// weights that meet wide activations (encoder, cross-attention K/V, prompt), so the GEMM inner loop carries
// no dequantization VALU work; the mat-vec path keeps streaming the quantized planes.
// -------------------------------------------------------------------------------------------------
template <int AT>
__global__ void __launch_bounds__(256) k_dequant_f16(const GemmArgs a) {
    const int64_t g = (int64_t) blockIdx.x * 256 + threadIdx.x;          // one 32-element group per thread
    const int kg = a.K >> 5;
    if (g >= (int64_t) a.M * kg) return;
    const int m = (int) (g / kg), k = (int) (g % kg) * 32;
    a_regs<AT> r;
    a_load<AT>(r, a, m, k, true);
    uint4 out[4];
    a_unpack<AT>(r, k, out);
    uint4 * d = (uint4 *) (a.dst + ((int64_t) m * a.K + k) * 2);
    #pragma unroll
    for (int i = 0; i < 4; i++) d[i] = out[i];
}

extern "C" int mi355x_dequant_f16(mi355x_ctx * ctx, const mi355x_tensor * A, void * dst) {
    const int K = (int) A->ne[0], M = (int) A->ne[1];
    if (!mi355x_type_is_quantized(A->type) || !t_is_contiguous(A) || A->ne[2] != 1 || A->ne[3] != 1 || K % type_block(A->type) ||
        ((uintptr_t) A->data % 16) || ((uintptr_t) dst % 16) || M <= 0 || K <= 0) return MI355X_E_UNSUPPORTED;
    GemmArgs k; memset(&k, 0, sizeof(k));
    k.A = (const char *) A->data; k.M = M; k.K = K; k.nbt = (int64_t) M * (K / type_block(A->type)); k.dst = (char *) dst;
    const int64_t ngroups = (int64_t) M * (K / 32);
    const dim3 g((uint32_t) ((ngroups + 255) / 256)), b(256);
    const double bytes = (double) mi355x_type_row_bytes(A->type, K) * M + (double) M * K * 2;
    int rc;
    switch (A->type) {
        case MI355X_TYPE_Q4_0: rc = emit(ctx, "dequant_f16", k_dequant_f16<MI355X_TYPE_Q4_0>, g, b, 0, k, bytes, 0); break;
        case MI355X_TYPE_Q5_0: rc = emit(ctx, "dequant_f16", k_dequant_f16<MI355X_TYPE_Q5_0>, g, b, 0, k, bytes, 0); break;
        case MI355X_TYPE_Q8_0: rc = emit(ctx, "dequant_f16", k_dequant_f16<MI355X_TYPE_Q8_0>, g, b, 0, k, bytes, 0); break;
        case MI355X_TYPE_Q4_K: rc = emit(ctx, "dequant_f16", k_dequant_f16<MI355X_TYPE_Q4_K>, g, b, 0, k, bytes, 0); break;
        default: rc = MI355X_E_UNSUPPORTED;
    }
    return rc;
}

// epilogue shared by the GEMM kernels: bias / scale / GELU / residual / f16 store on the 2 x NT accumulator tiles of a wave
template <int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs & a, floatx16 (&acc)[2][NT], int m0, int64_t n0, int wm, int wn, int WN, int lane) {
    // epilogue: C[i = A row][j = B row]: lane holds column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    // (F16 destinations — the K / V projections whose ggml_cpy into an F16 tensor is folded in — take the vector path too since r03:
    //  four halves per 8-byte store instead of four 2-byte stores)
    const bool vec_ok = (a.M % 4 == 0) && ((uintptr_t) a.dst % 16 == 0) && (a.dst_nb1 % 16 == 0) &&
                        (!a.residual || (((uintptr_t) a.residual % 16 == 0) && (a.res_nb1 % 16 == 0))) && (!a.bias || a.bias_t || ((uintptr_t) a.bias % 16 == 0));
    if (a.prep) {
        // The result is the activation matrix of the NEXT GEMM (fc1 + GELU -> fc2, src/whisper.cpp:2224-2238): write what k_prep_act
        // (mode 1) would make of it — the reference's Q8_0 rounding of every 32 consecutive features of a token, stored as f16(d*q) —
        // straight from the accumulators.  A 32-row block of one token is the 16 registers of this lane and of lane^32.
        // Host guarantees M % 32 == 0 and the 16-byte alignments of the vector path.
        #pragma unroll
        for (int i = 0; i < 2; i++) {
            #pragma unroll
            for (int j = 0; j < NT; j++) {
                const int64_t t = n0 + wn*WN + j*32 + (lane & 31);
                const int mb = m0 + wm*64 + i*32;
                if (t >= a.T || mb >= a.M) continue;            // the same for both lanes of a pair
                float v[16];
                float amax = 0.0f;
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int m = mb + 8*g + 4*(lane >> 5);
                    float x[4] = { acc[i][j][4*g], acc[i][j][4*g+1], acc[i][j][4*g+2], acc[i][j][4*g+3] };
                    if (a.bias) {
                        if (a.bias_t) { const float b = a.bias[t]; x[0] += b; x[1] += b; x[2] += b; x[3] += b; }
                        else { const float4 b = *(const float4 *) (a.bias + m); x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w; }
                    }
                    if (a.has_scale) { x[0] *= a.scale; x[1] *= a.scale; x[2] *= a.scale; x[3] *= a.scale; }
                    if (a.gelu) { x[0] = gelu_lut(x[0], a.gelu_tab); x[1] = gelu_lut(x[1], a.gelu_tab); x[2] = gelu_lut(x[2], a.gelu_tab); x[3] = gelu_lut(x[3], a.gelu_tab); }
                    if (a.residual) { const float4 r4 = *(const float4 *) (a.residual + t*a.res_nb1 + (int64_t) m*4); x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w; }
                    if (!a.prep_only) *(float4 *) (a.dst + t*a.dst_nb1 + (int64_t) m*4) = make_float4(x[0], x[1], x[2], x[3]);
                    #pragma unroll
                    for (int e = 0; e < 4; e++) { v[4*g + e] = x[e]; amax = fmaxf(amax, fabsf(x[e])); }
                }
                amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                const float d  = round_f16(amax / 127.0f);
                const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int m = mb + 8*g + 4*(lane >> 5);
                    float q[4];
                    #pragma unroll
                    for (int e = 0; e < 4; e++) q[e] = fminf(fmaxf(d * rintf(v[4*g + e]*id), -65504.0f), 65504.0f);
                    *(uint2 *) (a.prep + t*a.M + m) = make_uint2(f2h(q[0]) | ((uint32_t) f2h(q[1]) << 16), f2h(q[2]) | ((uint32_t) f2h(q[3]) << 16));
                }
            }
        }
        return;
    }
    #pragma unroll
    for (int i = 0; i < 2; i++) {
        #pragma unroll
        for (int j = 0; j < NT; j++) {
            const int64_t t = n0 + wn*WN + j*32 + (lane & 31);
            if (t >= a.T) continue;
            #pragma unroll
            for (int g = 0; g < 4; g++) {
                const int m = m0 + wm*64 + i*32 + 8*g + 4*(lane >> 5);
                if (m >= a.M) continue;
                float v[4] = { acc[i][j][4*g], acc[i][j][4*g+1], acc[i][j][4*g+2], acc[i][j][4*g+3] };
                if (vec_ok) {                                   // m % 4 == 0 and M % 4 == 0  =>  m+3 < M
                    if (a.bias) {
                        if (a.bias_t) { const float b = a.bias[t]; v[0] += b; v[1] += b; v[2] += b; v[3] += b; }
                        else { const float4 b = *(const float4 *) (a.bias + m); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                    }
                    if (a.has_scale) { v[0] *= a.scale; v[1] *= a.scale; v[2] *= a.scale; v[3] *= a.scale; }
                    if (a.gelu) { v[0] = gelu_lut(v[0], a.gelu_tab); v[1] = gelu_lut(v[1], a.gelu_tab); v[2] = gelu_lut(v[2], a.gelu_tab); v[3] = gelu_lut(v[3], a.gelu_tab); }
                    if (a.residual) { const float4 r4 = *(const float4 *) (a.residual + t*a.res_nb1 + (int64_t) m*4); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                    if (a.dst_f16) *(uint2 *) (a.dst + t*a.dst_nb1 + (int64_t) m*2) = make_uint2(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]));
                    else           *(float4 *) (a.dst + t*a.dst_nb1 + (int64_t) m*4) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    #pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (m + e >= a.M) break;
                        float x = v[e];
                        if (a.bias) x += a.bias[a.bias_t ? t : (int64_t) (m + e)];
                        if (a.has_scale) x *= a.scale;
                        if (a.gelu) x = gelu_lut(x, a.gelu_tab);
                        if (a.residual) x += *(const float *) (a.residual + t*a.res_nb1 + (int64_t) (m + e)*4);
                        if (a.dst_f16) *(uint16_t *) (a.dst + t*a.dst_nb1 + (int64_t) (m + e)*2) = f2h(x);
                        else           *(float *)    (a.dst + t*a.dst_nb1 + (int64_t) (m + e)*4) = x;
                    }
                }
            }
        }
    }
}

// The same epilogue through a wave-private LDS transposition (round 6).  The MFMA's C layout gives a lane ONE token and 16 features in four runs of 4:
// stored straight from the registers, a wave instruction touches 32 token rows with 32 contiguous bytes each (and reads the residual the same way) —
// the stores, not the products, were 20-30 us of a 45-55 us encoder product (profiles/r06_gemm_dq_ablation.txt).  Here a wave writes each 32-feature strip
// of its piece into its own [tokens][36 floats] LDS region and reads it back with 8 lanes per token row: bias, residual, destination and the next GEMM's
// activations all move as whole 128-byte (64-byte for f16) lines, and a 32-feature quantization block of a token is 8 neighbouring lanes (three shuffles).
// Per element the operations and their order are gemm_epilogue's: results are bit-identical.  Needs M % 4 == 0 and 16-byte aligned operands (caller checks).
// `tw`: this wave's region, NT * 32 * 36 floats; LDS operations of one wave execute in order, so no barrier separates the writes from the reads.
__device__ __forceinline__ bool gemm_epilogue_vec_ok(const GemmArgs & a) {
    return (a.M % 4 == 0) && ((uintptr_t) a.dst % 16 == 0) && (a.dst_nb1 % 16 == 0) &&
           (!a.residual || (((uintptr_t) a.residual % 16 == 0) && (a.res_nb1 % 16 == 0))) && (!a.bias || a.bias_t || ((uintptr_t) a.bias % 16 == 0));
}
__device__ __forceinline__ float gelu_win(float x, const uint16_t * __restrict__ win, const uint16_t * __restrict__ tab);
template <int NT>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs & a, floatx16 (&acc)[2][NT], int m0, int64_t n0, int wm, int wn, int WN, int lane, float * tw, const uint16_t * gelu_window = nullptr) {
    const int l31 = lane & 31, hf = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    // the residual rows of both strips are requested before anything else: the loads may alias the stores below (an in-place residual is allowed), so
    // left inside the loop every one of them waits for its own round trip to memory (16 dependent trips = the 16 k cycles of the fc2 / O epilogues)
    float4 res[2][NT*4];
    const int abl = a.ablate;
    if (a.residual && !(abl & 256)) {
        #pragma unroll
        for (int i = 0; i < 2; i++) {
            const int m = m0 + wm*64 + i*32 + c4;
            #pragma unroll
            for (int it = 0; it < NT*4; it++) {
                const int64_t t = n0 + wn*WN + it*8 + rr;
                res[i][it] = (m < a.M && t < a.T) ? *(const float4 *) (a.residual + t*a.res_nb1 + (int64_t) m*4) : make_float4(0, 0, 0, 0);
            }
        }
    }
    #pragma unroll
    for (int i = 0; i < 2; i++) {
        const int mb = m0 + wm*64 + i*32;
        if (mb >= a.M) continue;                                    // wave-uniform
        #pragma unroll
        for (int j = 0; j < NT; j++)
            #pragma unroll
            for (int g = 0; g < 4; g++)
                *(float4 *) (tw + (j*32 + l31)*36 + 8*g + 4*hf) = make_float4(acc[i][j][4*g], acc[i][j][4*g+1], acc[i][j][4*g+2], acc[i][j][4*g+3]);
        const int m = mb + c4;
        const bool m_ok = m < a.M;                                  // M % 4 == 0  =>  m + 3 < M
        float4 b4 = make_float4(0, 0, 0, 0);
        if (a.bias && !a.bias_t && m_ok) b4 = *(const float4 *) (a.bias + m);
        if (abl & 512) continue;
        #pragma unroll
        for (int it = 0; it < NT*4; it++) {
            const int row = it*8 + rr;
            const int64_t t = n0 + wn*WN + row;
            const float4 v4 = *(const float4 *) (tw + row*36 + c4);
            const bool ok = m_ok && t < a.T;
            float x[4] = { v4.x, v4.y, v4.z, v4.w };
            if (ok) {
                if (a.bias) {
                    if (a.bias_t) { const float b = a.bias[t]; x[0] += b; x[1] += b; x[2] += b; x[3] += b; }
                    else { x[0] += b4.x; x[1] += b4.y; x[2] += b4.z; x[3] += b4.w; }
                }
                if (a.has_scale) { x[0] *= a.scale; x[1] *= a.scale; x[2] *= a.scale; x[3] *= a.scale; }
                if (a.gelu && !(abl & 32)) {
                    if (gelu_window) { x[0] = gelu_win(x[0], gelu_window, a.gelu_tab); x[1] = gelu_win(x[1], gelu_window, a.gelu_tab); x[2] = gelu_win(x[2], gelu_window, a.gelu_tab); x[3] = gelu_win(x[3], gelu_window, a.gelu_tab); }
                    else { x[0] = gelu_lut(x[0], a.gelu_tab); x[1] = gelu_lut(x[1], a.gelu_tab); x[2] = gelu_lut(x[2], a.gelu_tab); x[3] = gelu_lut(x[3], a.gelu_tab); }
                }
                if (a.residual && !(abl & 256)) { const float4 r4 = res[i][it]; x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w; }
                if (!a.prep_only && !(abl & 128)) {
                    if (a.dst_f16) *(uint2 *) (a.dst + t*a.dst_nb1 + (int64_t) m*2) = make_uint2(pack_h2(x[0], x[1]), pack_h2(x[2], x[3]));
                    else           *(float4 *) (a.dst + t*a.dst_nb1 + (int64_t) m*4) = make_float4(x[0], x[1], x[2], x[3]);
                }
            }
            if (a.prep && !(abl & 64)) {
                // k_prep_act (mode 1) of the result: the 32 features of this token's block are this lane's four and those of the 7 lanes beside it
                // (host: M % 32 == 0, so a block is complete or absent)
                float amax = ok ? fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))) : 0.0f;
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                if (ok) {
                    const float d  = round_f16(amax / 127.0f);
                    const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
                    float q[4];
                    #pragma unroll
                    for (int e = 0; e < 4; e++) q[e] = fminf(fmaxf(d * rintf(x[e]*id), -65504.0f), 65504.0f);
                    *(uint2 *) (a.prep + t*a.M + m) = make_uint2(pack_h2(q[0], q[1]), pack_h2(q[2], q[3]));
                }
            }
        }
    }
}

template <int AT, int BN>
__global__ void __launch_bounds__(256) k_gemm_mfma(const GemmArgs a) {
    constexpr int WN = BN / 2;            // wave tile width (tokens)
    constexpr int NT = WN / 32;           // 32x32 tiles per wave along n
    constexpr int BSLOTS = BN * 8 / 256;  // 16-byte B slots per thread (4 or 2)
    __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * 128];
    char * ldsA = lds; char * ldsB = lds + BM*128;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int64_t n0 = (int64_t) blockIdx.y * BN;

    // staging assignments
    const int arow = tid >> 1, ahalf = tid & 1;
    const bool a_valid = (m0 + arow) < a.M;
    const int brow = BN == 128 ? (tid >> 1) : (tid >> 2);
    const int bslot0 = BN == 128 ? (tid & 1) * 4 : (tid & 3) * 2;
    const bool b_valid = (n0 + brow) < a.T;
    const uint16_t * bptr = a.B + (n0 + brow) * a.ldb;

    a_regs<AT> ar;
    uint4 br[BSLOTS];

    auto load_tile = [&](int k0) {
        a_load<AT>(ar, a, m0 + arow, k0 + ahalf*32, a_valid);
        #pragma unroll
        for (int i = 0; i < BSLOTS; i++) {
            const int k = k0 + (bslot0 + i)*8;
            br[i] = (b_valid && k < a.K) ? *(const uint4 *) (bptr + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int k0) {
        a_store<AT>(ar, ldsA, arow, ahalf, k0 + ahalf*32);
        #pragma unroll
        for (int i = 0; i < BSLOTS; i++) *(uint4 *) (ldsB + lds_off(brow, bslot0 + i)) = br[i];
    };

    floatx16 acc[2][NT];
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int j = 0; j < NT; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int nk = (a.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);

        #pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            half8_t af[2], bf[NT];
            const int slot = kk*2 + (lane >> 5);
            #pragma unroll
            for (int i = 0; i < 2; i++) af[i] = *(const half8_t *) (ldsA + lds_off(wm*64 + i*32 + (lane & 31), slot));
            #pragma unroll
            for (int j = 0; j < NT; j++) bf[j] = *(const half8_t *) (ldsB + lds_off(wn*WN + j*32 + (lane & 31), slot));
            #pragma unroll
            for (int i = 0; i < 2; i++)
                #pragma unroll
                for (int j = 0; j < NT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) { store_tile((kt + 1) * BK); __syncthreads(); }
    }

    gemm_epilogue<NT>(a, acc, m0, n0, wm, wn, WN, lane);
}

// -------------------------------------------------------------------------------------------------
// k_gemm_f16_ring: the GEMM for two plain f16 operands, A [M][K] (a weight's f16 copy, or an f16 weight) and B [T][K]
// (prepared activations), both K-contiguous.  Nothing passes through VGPRs on the way in: every wave fills its share of a
// ring of NST LDS stages with `global_load_lds` (16 bytes per lane, 1 KB = 8 rows x 128 B per instruction; the XOR swizzle
// that keeps the ds_read_b128 fragment reads conflict-free is applied to the per-lane GLOBAL address because the LDS side
// of an LDS-DMA is lane-linear), NST-1 K-steps ahead of the MFMAs.  One raw s_barrier per K-step; the wait for a stage is a
// counted s_waitcnt vmcnt(G * stages_still_in_flight), never 0 inside the loop.  All LDS is ONE array and no ordinary
// global load is issued inside the loop (either would make hipcc drain the DMA queue every step).  Same fragment
// layout, MFMA and K order as k_gemm_mfma => bit-identical results.
// -------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// TM = rows of A per tile: 128 (4 waves, 2 x 2) or 256 (8 waves, 4 x 2: r03 — twice the arithmetic per byte the LDS-DMA has to
// deliver, two waves per SIMD inside ONE workgroup; for products with enough 256-row tiles to cover the chip).  Every wave owns a
// 64 x BN/2 piece whatever TM is, so fragments, MFMA order and epilogue are the same code: results are bit-identical across TM.
template <int TM, int BN, int NST>
__device__ __forceinline__ void gemm_ring_tile(const GemmArgs & a, const int tile, char * ring) {
    constexpr int NW = TM / 32;                       // waves per workgroup
    constexpr int WN = BN / 2, NT = WN / 32;
    constexpr int STAGE = (TM + BN) * 128;            // bytes per stage: A tile [TM][64] f16, B tile [BN][64] f16
    constexpr int GA = TM / NW / 8, GB = BN / NW / 8; // LDS-DMA instructions per wave and stage: 8 rows each
    constexpr int G = GA + GB;
    static_assert(GA >= 1 && GB >= 1, "every wave moves at least one 8-row group of each operand");

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2, so
    // with the natural order each XCD ends up reading (nearly) all of A and all of B: PMC showed 55 MB of fabric reads for
    // the 7 MB of operands of a 1280 x 1500 x 1280 product.  Here XCD x = id % 8 works through the contiguous tile range
    // [x*per, (x+1)*per) of an order in which neighbours share an operand (whole columns of tiles, or whole rows —
    // whichever moves fewer bytes, chosen on the host).
    const int mi = a.m_major ? tile / a.nt : tile % a.mt;
    const int ni = a.m_major ? tile % a.nt : tile / a.mt;
    const int m0 = mi * TM;
    const int64_t n0 = (int64_t) ni * BN;
    const int nk = a.K / BK;

    // this lane's share of a stage: row (lane >> 3) of each 8-row group, physical 16-byte slot (lane & 7), which holds the
    // logical k-slot (lane & 7) ^ ((row >> 1) & 7)  (lds_off).  Rows past the matrix edge are clamped (their columns /
    // rows are never stored).
    const char * ga[GA]; const char * gb[GB];
    #pragma unroll
    for (int i = 0; i < GA; i++) {
        const int r = wave * (TM / NW) + i * 8 + (lane >> 3);
        const int mr = m0 + r < a.M ? m0 + r : a.M - 1;
        ga[i] = a.A + (int64_t) mr * a.a_nb1 + ((((lane & 7) ^ ((r >> 1) & 7))) << 4);
    }
    #pragma unroll
    for (int i = 0; i < GB; i++) {
        const int r = wave * (BN / NW) + i * 8 + (lane >> 3);
        const int64_t tr = n0 + r < a.T ? n0 + r : a.T - 1;
        gb[i] = (const char *) (a.B + tr * a.ldb) + ((((lane & 7) ^ ((r >> 1) & 7))) << 4);
    }
    auto issue = [&](int kt) {
        char * st = ring + (kt % NST) * STAGE;
        const int koff = kt * BK * 2;
        #pragma unroll
        for (int i = 0; i < GA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (ga[i] + koff),
                                             (__attribute__((address_space(3))) void *) (st + (wave * (TM / NW) + i * 8) * 128), 16, 0, 0);
        #pragma unroll
        for (int i = 0; i < GB; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (gb[i] + koff),
                                             (__attribute__((address_space(3))) void *) (st + TM*128 + (wave * (BN / NW) + i * 8) * 128), 16, 0, 0);
    };

    floatx16 acc[2][NT];
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int j = 0; j < NT; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    #pragma unroll
    for (int p = 0; p < NST - 1; p++) if (p < nk) issue(p);

    for (int kt = 0; kt < nk; kt++) {
        // stages issued after stage kt and still allowed in flight: min(NST - 2, nk - 1 - kt)
        const int ahead = nk - 1 - kt;
        if (ahead >= NST - 2)      wait_vmcnt<G * (NST - 2)>();
        else if (ahead == 1)       wait_vmcnt<G>();
        else                       wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                 // stage kt complete in LDS for every wave; everyone is done reading stage kt-1
        if (kt + NST - 1 < nk) issue(kt + NST - 1);    // refill the slot of stage kt-1

        const char * ldsA = ring + (kt % NST) * STAGE;
        const char * ldsB = ldsA + TM*128;
        // (r03: requesting the fragment reads two slices ahead of their MFMAs, in registers of their own and with the order pinned by
        // sched_barrier, moved nothing — 32.3 vs 31.4 us for the fc1 product, profiles/r03b_gemm_pipelined_v1.txt: a K-step is bound by
        // what the LDS-DMA delivers, see DESIGN.md §5 — so the slice loop is left to the compiler)
        #pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            half8_t af[2], bf[NT];
            const int slot = kk*2 + (lane >> 5);
            #pragma unroll
            for (int i = 0; i < 2; i++) af[i] = *(const half8_t *) (ldsA + lds_off(wm*64 + i*32 + (lane & 31), slot));
            #pragma unroll
            for (int j = 0; j < NT; j++) bf[j] = *(const half8_t *) (ldsB + lds_off(wn*WN + j*32 + (lane & 31), slot));
            #pragma unroll
            for (int i = 0; i < 2; i++)
                #pragma unroll
                for (int j = 0; j < NT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    gemm_epilogue<NT>(a, acc, m0, n0, wm, wn, WN, lane);
}

template <int BN, int NST, int TM>
__global__ void __launch_bounds__(TM * 2) k_gemm_f16_ring(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = xcd * a.per + idx;
    if (tile >= a.mt * a.nt) return;                  // padding blocks of the last XCD range (uniform exit, before any barrier)
    gemm_ring_tile<TM, BN, NST>(a, tile, ring);
}

// Grouped form: up to GEMM_GROUP_MAX independent products with the SAME activation matrix B and the same shape (the Q / K / V
// projections of an encoder layer; the cross-attention K / V projections of consecutive decoder layers), one launch.  The tile
// space is the concatenation of the members' tile spaces, dealt to the XCDs in contiguous ranges like the single form, so an XCD
// mostly stays inside one member (one A) and B is shared by all of them.  Three 1280 x 1500 x 1280 products alone are 3 x 240
// tiles of 128 x 64 on 256 CUs with three launch / ring-fill / drain phases; together they are 360 tiles of 128 x 128 with one.
// Every tile runs exactly the code of the single form: results are bit-identical.
#define GEMM_GROUP_MAX 8
struct GemmGroupArgs { GemmArgs g[GEMM_GROUP_MAX]; int n, tiles_per_member, per; };

template <int BN, int NST, int TM>
__global__ void __launch_bounds__(TM * 2) k_gemm_f16_ring_group(const GemmGroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t = xcd * ga.per + idx;
    if (t >= ga.n * ga.tiles_per_member) return;
    const int gi = t / ga.tiles_per_member;
    gemm_ring_tile<TM, BN, NST>(ga.g[gi], t - gi * ga.tiles_per_member, ring);
}

// tile counts of one member + the XCD-aware tile order for ranges of `per` consecutive tiles
template <int TM, int BN>
static void ring_tiling(GemmArgs & k, int64_t per) {
    k.mt = (int) ((k.M + TM - 1) / TM); k.nt = (int) ((k.T + BN - 1) / BN);
    const int64_t ntiles = (int64_t) k.mt * k.nt;
    k.per = (int) (per < ntiles ? per : ntiles);
    // bytes each XCD pulls through its L2 for its `per` consecutive tiles, column-major vs row-major tile order
    const double a_tile = (double) TM * k.K * 2, b_tile = (double) BN * k.K * 2;
    const double col_major = a_tile * (k.per < k.mt ? k.per : k.mt) + b_tile * ((k.per + k.mt - 1) / k.mt + (k.per % k.mt ? 1 : 0));
    const double row_major = b_tile * (k.per < k.nt ? k.per : k.nt) + a_tile * ((k.per + k.nt - 1) / k.nt + (k.per % k.nt ? 1 : 0));
    k.m_major = row_major < col_major ? 1 : 0;
}

template <typename F>
static int ring_lds_attr(mi355x_ctx * ctx, F func, uint32_t lds, std::atomic<bool> * attr_set) {
    const int dev = ctx->device & 63;                   // > 64 KB of dynamic LDS needs the attribute once per function AND device
    if (!attr_set[dev].load()) {
        if (hipFuncSetAttribute((const void *) func, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) {
            (void) hipGetLastError();
            return MI355X_E_UNSUPPORTED;
        }
        attr_set[dev].store(true);
    }
    return 0;
}

template <int BN, int NST, int TM = BM>
static int launch_ring(mi355x_ctx * ctx, const GemmArgs & k0, double bytes, double flops) {
    constexpr uint32_t lds = (uint32_t) NST * (TM + BN) * 128;
    GemmArgs k = k0;
    const int64_t ntiles = ((k.M + TM - 1) / TM) * ((k.T + BN - 1) / BN);
    ring_tiling<TM, BN>(k, (ntiles + 7) / 8);
    k.per = (int) ((ntiles + 7) / 8);
    static std::atomic<bool> attr_set[64];
    if (ring_lds_attr(ctx, k_gemm_f16_ring<BN, NST, TM>, lds, attr_set) != 0) return MI355X_E_UNSUPPORTED;
    return emit(ctx, "gemm_f16_ring", k_gemm_f16_ring<BN, NST, TM>, dim3((uint32_t) (8 * k.per)), dim3(TM * 2), lds, k, bytes, flops);
}

template <int BN, int NST, int TM = BM>
static int launch_ring_group(mi355x_ctx * ctx, const GemmArgs * members, int n, double bytes, double flops) {
    constexpr uint32_t lds = (uint32_t) NST * (TM + BN) * 128;
    GemmGroupArgs ga; memset(&ga, 0, sizeof(ga));
    const int64_t tiles = ((members[0].M + TM - 1) / TM) * ((members[0].T + BN - 1) / BN);
    const int64_t per = (n * tiles + 7) / 8;
    for (int i = 0; i < n; i++) { ga.g[i] = members[i]; ring_tiling<TM, BN>(ga.g[i], per); }
    ga.n = n; ga.tiles_per_member = (int) tiles; ga.per = (int) per;
    static std::atomic<bool> attr_set[64];
    if (ring_lds_attr(ctx, k_gemm_f16_ring_group<BN, NST, TM>, lds, attr_set) != 0) return MI355X_E_UNSUPPORTED;
    return emit(ctx, "gemm_f16_ring_group", k_gemm_f16_ring_group<BN, NST, TM>, dim3((uint32_t) (8 * per)), dim3(TM * 2), lds, ga, bytes, flops);
}

// -------------------------------------------------------------------------------------------------
// k_gemm_dq (round 6): the product of a QUANTIZED weight with wide activations — "per-warp dequant into LDS tiles feeding MFMA" without an
// f16 copy of the weight anywhere.  What bounds a K-step of these products on a CU is not the matrix pipe but the vector-memory path that
// fills LDS (64 B/clk per CU at best): a 128 x 128 tile of two f16 operands needs 32 KB per 64 MFMAs (= 512 matrix-pipe cycles per SIMD:
// exactly the path's peak — the ring kernel above runs at 28-34 % of the MFMA rate, r03b_gemm_fixed_cost.txt).  So
//   * A travels in its block-quantized planar form (0.56 - 1.06 B/weight instead of 2): the planes of a K-step reach LDS as they lie in memory, by LDS-DMA
//     ("raw" stage, three deep), and waves 0..3 — one per SIMD — unpack one (row, 32-element block) pair per thread to f16 with packed f16 arithmetic
//     (a_unpack_fast: the values of mi355x_dequant_f16, bit for bit) into the XOR-swizzled A stage that the NEXT step reads, beside the MFMAs of the SIMD's other wave;
//   * B (prepared f16 activations [T][K]) arrives by LDS-DMA, 8 rows x 128 B per instruction, swizzle on the global side (as the ring), three stages deep;
//   * every operand being a DMA, every wait is an exact vmcnt count (an ordinary register load beside a DMA makes hipcc drain the whole queue: versions 1 and 2);
//   * the tile is 128 (weight rows) x BN tokens with BN = 256 where the product has enough tiles (8 waves, 2 x 4; 37.6 KB per 128 MFMAs for Q5_0) and 128
//     otherwise (4 waves); every wave owns a 64 x 64 piece: fragments, MFMA order and per-element epilogue operations are the ring kernel's, so results are
//     bit-identical to it on the weight's f16 copy (tests/test_gpu_encoder.py);
//   * the epilogue goes through a wave-private LDS transposition (gemm_epilogue_lds above) with the GELU table in LDS.
// Measured: correct and SLOWER than the ring on the copy — the row-scattered planes cost more cache-line requests than the copy's contiguous rows, and the vector-memory
// path of a CU serves ~0.2 of them per clock whatever they carry (profiles/r06_gemm_dq_anatomy.txt).  Opt-in: GGML_MI355X_MMQ=3 / MI355X_OPT_DQ_GEMM.
// ONE barrier per K-step: wait (all but the newest DMA batch + own A stores) -> barrier -> request raw A(k+3), B(k+2) -> unpack A(k+1) -> MFMAs(k).
// (-DMI355X_RING_ONLY leaves the family out: tests/test_host.py's ISA check of the ring kernels compiles this file on the CPU suite's clock.)
// -------------------------------------------------------------------------------------------------
#ifndef MI355X_RING_ONLY
// ---- unpack of one 32-element block to f16 without a float in sight ----------------------------------------------------------------
// a_unpack computes f16(float(q) * d) per weight (extract, int -> float, multiply, float -> f16: 5-6 VALU operations per weight, ~1000 cycles per
// K-step of a staging wave, profiles/r06_gemm_dq_anatomy.txt).  The same values from packed f16 arithmetic: a 4..8-bit unsigned u placed in the
// mantissa of 1024.0 (bits 0x6400 | u) IS the f16 number 1024 + u; subtracting 1024 + offset is exact; one v_pk_mul_f16 by d rounds the product
// (exact in f32: <= 8 x 11 bits) once, to nearest even, denormals kept — f16(float(q) * d) bit for bit (tests: k_gemm_dq == k_gemm_mfma == ring
// on mi355x_dequant_f16's copy).  Two weights per v_perm_b32 / v_pk_add_f16 / v_pk_mul_f16.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t h2_bits(half2_t v) { return __builtin_bit_cast(uint32_t, v); }
// bytes b0..b3 of x (each < 256) -> (0x6400 | b0, 0x6400 | b1) and (0x6400 | b2, 0x6400 | b3), minus `off`, times d
__device__ __forceinline__ void dq_pairs(uint32_t x, half2_t off, half2_t d2, uint32_t & o01, uint32_t & o23) {
    const uint32_t p01 = __builtin_amdgcn_perm(0x64646464u, x, 0x04010400u);
    const uint32_t p23 = __builtin_amdgcn_perm(0x64646464u, x, 0x04030402u);
    o01 = h2_bits((__builtin_bit_cast(half2_t, p01) - off) * d2);
    o23 = h2_bits((__builtin_bit_cast(half2_t, p23) - off) * d2);
}
template <int AT>
__device__ __forceinline__ void a_unpack_fast(const a_regs<AT> & r, int k, uint4 (&out)[4], const uint32_t * lut) {
    if constexpr (AT == MI355X_TYPE_Q8_0) {
        const half_t dh = __builtin_bit_cast(half_t, r.d);
        const half2_t d2 = { dh, dh }, off = { (half_t) 1152.0f, (half_t) 1152.0f };          // u = q + 128
        const uint32_t w[8] = { r.q0.x, r.q0.y, r.q0.z, r.q0.w, r.q1.x, r.q1.y, r.q1.z, r.q1.w };
        uint32_t o[16];
        #pragma unroll
        for (int i = 0; i < 8; i++) dq_pairs(w[i] ^ 0x80808080u, off, d2, o[2*i], o[2*i+1]);
        #pragma unroll
        for (int i = 0; i < 4; i++) out[i] = make_uint4(o[4*i], o[4*i+1], o[4*i+2], o[4*i+3]);
    } else if constexpr (AT == MI355X_TYPE_Q4_0 || AT == MI355X_TYPE_Q5_0) {
        const half_t dh = __builtin_bit_cast(half_t, r.d);
        const half2_t d2 = { dh, dh };
        const half2_t off = AT == MI355X_TYPE_Q5_0 ? half2_t{ (half_t) 1040.0f, (half_t) 1040.0f } : half2_t{ (half_t) 1032.0f, (half_t) 1032.0f };
        const uint32_t w[4] = { r.q.x, r.q.y, r.q.z, r.q.w };
        uint32_t o[16];
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t lo = w[i] & 0x0F0F0F0Fu, hi = (w[i] >> 4) & 0x0F0F0F0Fu;                   // elements 4i .. 4i+3 and 16+4i .. 16+4i+3
            if constexpr (AT == MI355X_TYPE_Q5_0) { lo |= lut[(r.qh >> (4*i)) & 0xFu]; hi |= lut[(r.qh >> (16 + 4*i)) & 0xFu]; }      // the fifth bits (table: bit k -> 0x10 in byte k)
            dq_pairs(lo, off, d2, o[2*i], o[2*i+1]);
            dq_pairs(hi, off, d2, o[8 + 2*i], o[8 + 2*i+1]);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) out[i] = make_uint4(o[4*i], o[4*i+1], o[4*i+2], o[4*i+3]);
    } else a_unpack<AT>(r, k, out);                    // Q4_K: d * sc * q - dmin * m is an f32 fma (kept as it is)
}

// f16-table GELU with the table in LDS: the entries of every f16 value with |x| < 16 (exponent fields 0 .. 18: 19 x 1024 values x 2 signs = 76 KB;
// |x| >= 10 never reaches the table: gelu_lut).  A wave's 64 look-ups in global memory are 64 different cache lines for the texture path
// (~64 cycles per instruction, and a dependent round trip each: the 36-45 k cycles of fc1's epilogue, profiles/r06_gemm_dq_anatomy.txt) and a few LDS cycles here.
#define GELU_WIN_N     (19 * 1024)
#define GELU_WIN_BYTES (2 * GELU_WIN_N * 2)
__device__ __forceinline__ float gelu_win(float x, const uint16_t * __restrict__ win, const uint16_t * __restrict__) {
    if (x <= -10.0f) return 0.0f;
    if (x >=  10.0f) return x;
    const uint32_t h = f2h(x);                         // |x| < 10: (h & 0x7FFF) < 0x4900 < GELU_WIN_N
    return h2f(win[(h & 0x7FFFu) + ((h >> 15) ? GELU_WIN_N : 0)]);
}

// ---- the weight planes of one K-step as they lie in memory, through LDS-DMA (the "raw" stage) --------------------------------------
// staging thread t = (tile row t >> 1, block t & 1 of the K-step), 256 of them (waves 0..3).  Every field of a_regs<AT> is a plane [256][field bytes]
// that the thread's OWN wave fills (lane-linear LDS destination = thread-linear plane); Q4_K: the super-block's dm / 12 scale bytes once per ROW
// ([128][4], [128][12]; rows 64 (w & 1) .. by wave w: waves 0 / 1 fetch dm, waves 2 / 3 the scales).  NA = LDS-DMA instructions per staging wave and K-step.
template <int AT> struct dq_raw;
// (only 4- and 16-byte DMA pieces: the two f16 scales of a row's K-step are ONE aligned dword — the blocks-per-row count is even — which both of the row's
//  threads fetch; the 12 scale bytes of a Q4_K super-block travel as 16, the four extra ones still inside the tensor: the dm plane follows the scales)
template <> struct dq_raw<MI355X_TYPE_Q4_0> { static constexpr int NA = 2, BYTES = 256 * 16 + 256 * 4; };
template <> struct dq_raw<MI355X_TYPE_Q5_0> { static constexpr int NA = 3, BYTES = 256 * 16 + 256 * 4 + 256 * 4; };
template <> struct dq_raw<MI355X_TYPE_Q8_0> { static constexpr int NA = 3, BYTES = 256 * 32 + 256 * 4; };
template <> struct dq_raw<MI355X_TYPE_Q4_K> { static constexpr int NA = 3, BYTES = 256 * 32 + 128 * 4 + 128 * 16; };
#define DQ_GLDS(gptr, lptr, SZ) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (gptr), (__attribute__((address_space(3))) void *) (lptr), SZ, 0, 0)

template <int AT>
__device__ __forceinline__ void dq_raw_issue(const GemmArgs & a, char * raw, int wave, int lane, int m0, int kt) {
    const int t = wave * 64 + lane, half = t & 1;
    const int am = m0 + (t >> 1) < a.M ? m0 + (t >> 1) : a.M - 1;          // rows past the edge are clamped (never stored)
    if constexpr (AT == MI355X_TYPE_Q4_K) {
        const qplanes<MI355X_TYPE_Q4_K> p(a.A, a.nbt);
        const int k = kt * BK + half * 32;                                   // both blocks of a K-step lie in ONE 64-element chunk: low / high nibbles of the same 32 bytes
        const int64_t sb = (int64_t) am * (a.K >> 8) + (k >> 8);
        const uint8_t * qs = p.qs + sb*128 + (((k & 255) >> 5) >> 1) * 32;
        DQ_GLDS(qs,      raw + wave * 1024,        16);
        DQ_GLDS(qs + 16, raw + 4096 + wave * 1024, 16);
        const int r = (wave & 1) * 64 + lane;
        const int rm = m0 + r < a.M ? m0 + r : a.M - 1;
        const int64_t sbr = (int64_t) rm * (a.K >> 8) + ((kt * BK) >> 8);
        if (wave < 2) DQ_GLDS(p.dm + sbr, raw + 8192 + (wave & 1) * 256, 4);
        else          DQ_GLDS(p.sc + sbr*12, raw + 8704 + (wave & 1) * 1024, 16);
    } else {
        const qplanes<AT> p(a.A, a.nbt);
        const int64_t ib = (int64_t) am * (a.K >> 5) + kt * 2 + half;
        const uint16_t * dpair = p.d + (ib - half);                           // both scales of the row's K-step (host: K % 64 == 0, so the pair is dword-aligned)
        if constexpr (AT == MI355X_TYPE_Q8_0) {
            DQ_GLDS(p.qs + ib*32,      raw + wave * 1024,        16);
            DQ_GLDS(p.qs + ib*32 + 16, raw + 4096 + wave * 1024, 16);
            DQ_GLDS(dpair,             raw + 8192 + wave * 256,  4);
        } else {
            DQ_GLDS(p.qs + ib*16, raw + wave * 1024, 16);
            if constexpr (AT == MI355X_TYPE_Q5_0) { DQ_GLDS(p.qh + ib, raw + 4096 + wave * 256, 4); DQ_GLDS(dpair, raw + 5120 + wave * 256, 4); }
            else DQ_GLDS(dpair, raw + 4096 + wave * 256, 4);
        }
    }
}
template <int AT>
__device__ __forceinline__ void dq_raw_read(a_regs<AT> & r, const char * raw, int t) {
    if constexpr (AT == MI355X_TYPE_Q4_K) {
        r.q0 = *(const uint4 *) (raw + t*16); r.q1 = *(const uint4 *) (raw + 4096 + t*16);
        r.dm = *(const uint32_t *) (raw + 8192 + (t >> 1)*4);
        const uint32_t * sc = (const uint32_t *) (raw + 8704 + (t >> 1)*16);
        r.sc[0] = sc[0]; r.sc[1] = sc[1]; r.sc[2] = sc[2];
    } else if constexpr (AT == MI355X_TYPE_Q8_0) {
        r.q0 = *(const uint4 *) (raw + t*16); r.q1 = *(const uint4 *) (raw + 4096 + t*16); r.d = (uint16_t) (*(const uint32_t *) (raw + 8192 + t*4) >> (16 * (t & 1)));
    } else if constexpr (AT == MI355X_TYPE_Q5_0) {
        r.q = *(const uint4 *) (raw + t*16); r.qh = *(const uint32_t *) (raw + 4096 + t*4); r.d = (uint16_t) (*(const uint32_t *) (raw + 5120 + t*4) >> (16 * (t & 1)));
    } else {
        r.q = *(const uint4 *) (raw + t*16); r.d = (uint16_t) (*(const uint32_t *) (raw + 4096 + t*4) >> (16 * (t & 1)));
    }
}

template <int N> __device__ __forceinline__ void dq_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory"); }

template <int AT, int BN>
__device__ __forceinline__ void gemm_dq_tile(const GemmArgs & a, const int tile, char * lds) {
    constexpr int NW = BN / 32;                        // waves: 4 or 8
    constexpr int WN = 64, NT = 2;
    constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128, NSB = 3, NSR = 3;
    constexpr int RAW = (dq_raw<AT>::BYTES + 255) & ~255, NA = dq_raw<AT>::NA;
    constexpr int OFF_B = 2 * A_STAGE;
    constexpr int OFF_RAW = OFF_B + NSB * B_STAGE;
    constexpr int OFF_LUT = OFF_RAW + NSR * RAW;       // Q5_0: 16 x u32
    constexpr int GB = BN / NW / 8;                    // LDS-DMA instructions per wave and stage of B (4)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const bool stager = tid < 256;                     // waves 0..3: one per SIMD
    const int mi = a.m_major ? tile / a.nt : tile % a.mt;
    const int ni = a.m_major ? tile % a.nt : tile / a.mt;
    const int m0 = mi * BM;
    const int64_t n0 = (int64_t) ni * BN;
    const int nk = a.K / BK;

    const int arow = (tid & 255) >> 1, ahalf = tid & 1;
    const char * gb[GB];
    #pragma unroll
    for (int i = 0; i < GB; i++) {
        const int r = wave * (BN / NW) + i * 8 + (lane >> 3);
        const int64_t tr = n0 + r < a.T ? n0 + r : a.T - 1;
        gb[i] = (const char *) (a.B + tr * a.ldb) + ((((lane & 7) ^ ((r >> 1) & 7))) << 4);
    }
    auto issue_b = [&](int kt) {
        char * st = lds + OFF_B + (kt % NSB) * B_STAGE;
        const int koff = kt * BK * 2;
        #pragma unroll
        for (int i = 0; i < GB; i++) DQ_GLDS(gb[i] + koff, st + (wave * (BN / NW) + i * 8) * 128, 16);
    };
    auto issue_a = [&](int kt) { dq_raw_issue<AT>(a, lds + OFF_RAW + (kt % NSR) * RAW, wave, lane, m0, kt); };
    const uint32_t * lut = (const uint32_t *) (lds + OFF_LUT);
    // raw planes of A(kt) -> f16 -> the A stage that step kt reads
    auto stage_a = [&](int kt) {
        a_regs<AT> r;
        dq_raw_read<AT>(r, lds + OFF_RAW + (kt % NSR) * RAW, tid);
        uint4 out[4];
        a_unpack_fast<AT>(r, kt * BK + ahalf*32, out, lut);
        char * st = lds + (kt & 1) * A_STAGE;
        #pragma unroll
        for (int i = 0; i < 4; i++) *(uint4 *) (st + lds_off(arow, ahalf*4 + i)) = out[i];
    };

    floatx16 acc[2][NT];
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int j = 0; j < NT; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int abl = a.ablate;
    // anatomy stamps: [0] entry, [1] loop entered, [2] loop left, [3] epilogue done, [4] cycles between the top of a step and its barrier's release (summed),
    // [5] staging section, [6] MFMA section — wave 0 of workgroup 0 in slots 0..7, its last wave in 8..15
    unsigned long long * dbg = (a.dbg && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == NW - 1)) ? a.dbg + (wave ? 8 : 0) : nullptr;
    unsigned long long t_wait = 0, t_stage = 0, t_mma = 0, ts = 0;
    if (dbg) dbg[0] = __builtin_amdgcn_s_memtime();
    if constexpr (AT == MI355X_TYPE_Q5_0) {
        if (tid < 16) ((uint32_t *) (lds + OFF_LUT))[tid] = ((tid & 1) ? 0x10u : 0u) | ((tid & 2) ? 0x1000u : 0u) | ((tid & 4) ? 0x100000u : 0u) | ((tid & 8) ? 0x10000000u : 0u);
    }
    // Every operand arrives by LDS-DMA, so every wait is an exact count (an ordinary register load beside a DMA makes hipcc drain the queue).  A wave's
    // requests in step s, in this order: raw A(s+3) [staging waves], B(s+2) — one "batch".  At the top of step s+1 everything but the newest batch has
    // landed: B(s+1) and raw A(s+2), which that step unpacks into the A stage step s+2 reads.  B and raw A are three stages deep, f16 A two.
    if (stager) issue_a(0);
    if (stager && nk > 1) issue_a(1);
    issue_b(0);
    if (stager && nk > 2) issue_a(2);
    if (nk > 1) issue_b(1);
    // the batch before the loop: raw A(2) + B(1) — raw A(0) (unpacked here), raw A(1), B(0) are older
    if (nk > 1) { if (stager && nk > 2) dq_wait<NA + GB>(); else dq_wait<GB>(); } else dq_wait<0>();
    __syncthreads();                                                       // (Q5_0: the bit table; Q4_K: the rows' dm / scales fetched by the neighbouring wave)
    if (stager) stage_a(0);

    if (dbg) dbg[1] = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; kt++) {
        if (dbg) ts = __builtin_amdgcn_s_memtime();
        // the batch of step kt-1: raw A(kt+2) [staging waves, if it exists] + B(kt+1) [if it exists]; own stores of A(kt) done (lgkmcnt)
        if (kt == 0) dq_wait<NA + GB + NA + GB>();                          // (nothing new to wait for: the prologue's wait covered B(0))
        else if (kt + 1 < nk) { if (stager && kt + 2 < nk) dq_wait<NA + GB>(); else dq_wait<GB>(); }
        else dq_wait<0>();
        __builtin_amdgcn_s_barrier();                                      // ... everybody's; and everybody has finished reading the stages of step kt-1
        if (dbg) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_wait += t - ts; ts = t; }
        if (stager && kt + 3 < nk && !(abl & 8)) issue_a(kt + 3);
        if (kt + 2 < nk && !(abl & 2)) issue_b(kt + 2);
        if (stager && kt + 1 < nk && !(abl & 1)) stage_a(kt + 1);
        if (dbg) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_stage += t - ts; ts = t; }
        const char * ldsA = lds + (kt & 1) * A_STAGE;
        const char * ldsB = lds + OFF_B + (kt % NSB) * B_STAGE;
        if (!(abl & 4))
        #pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            half8_t af[2], bf[NT];
            const int slot = kk*2 + (lane >> 5);
            #pragma unroll
            for (int i = 0; i < 2; i++) af[i] = *(const half8_t *) (ldsA + lds_off(wm*64 + i*32 + (lane & 31), slot));
            #pragma unroll
            for (int j = 0; j < NT; j++) bf[j] = *(const half8_t *) (ldsB + lds_off(wn*WN + j*32 + (lane & 31), slot));
            #pragma unroll
            for (int i = 0; i < 2; i++)
                #pragma unroll
                for (int j = 0; j < NT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (dbg) { asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][NT-1])); const unsigned long long t = __builtin_amdgcn_s_memtime(); t_mma += t - ts; }
    }
    if (dbg) { dbg[2] = __builtin_amdgcn_s_memtime(); dbg[4] = t_wait; dbg[5] = t_stage; dbg[6] = t_mma; }
    if (gemm_epilogue_vec_ok(a) && !(abl & 16)) {
        __syncthreads();                                                   // every wave has finished reading the last stage: the staging area becomes the transposition buffer
        const uint16_t * win = nullptr;
        if (a.gelu && a.lds_bytes >= NW * (NT*32*36*4) + GELU_WIN_BYTES) {
            // the GELU table's 2 x 38 KB behind the transposition regions
            uint16_t * wdst = (uint16_t *) (lds + NW * (NT*32*36*4));
            for (int e = tid; e < GELU_WIN_BYTES / 16; e += BN * 2) {
                const int neg = e >= GELU_WIN_N * 2 / 16;
                ((uint4 *) wdst)[e] = ((const uint4 *) (a.gelu_tab + (neg ? 0x8000 : 0)))[e - neg * (GELU_WIN_N * 2 / 16)];
            }
            __syncthreads();
            win = wdst;
        }
        gemm_epilogue_lds<NT>(a, acc, m0, n0, wm, wn, WN, lane, (float *) lds + wave * (NT*32*36), win);
    } else gemm_epilogue<NT>(a, acc, m0, n0, wm, wn, WN, lane);
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[3] = __builtin_amdgcn_s_memtime(); }
}

template <int AT, int BN>
__global__ void __launch_bounds__(BN * 2) k_gemm_dq(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dq_lds[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = xcd * a.per + idx;
    if (tile >= a.mt * a.nt) return;
    gemm_dq_tile<AT, BN>(a, tile, dq_lds);
}

template <int AT, int BN>
__global__ void __launch_bounds__(BN * 2) k_gemm_dq_group(const GemmGroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char dq_lds_g[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t = xcd * ga.per + idx;
    if (t >= ga.n * ga.tiles_per_member) return;
    const int gi = t / ga.tiles_per_member;
    gemm_dq_tile<AT, BN>(ga.g[gi], t - gi * ga.tiles_per_member, dq_lds_g);
}

// token-tile width: 256 when that still gives (nearly) every CU a tile, else 128 (test option MI355X_OPT_DQ_BN forces one)
static int dq_pick_bn(mi355x_ctx * ctx, const GemmArgs & k, int n) {
    const int force = mi355x_opt(MI355X_OPT_DQ_BN, 0);
    if (force == 128 || force == 256) return force;
    const int64_t t256 = (int64_t) n * ((k.M + BM - 1) / BM) * ((k.T + 255) / 256);
    return t256 * 10 >= (int64_t) ctx->n_cu * 6 ? 256 : 128;
}

template <int AT, int BN>
static int launch_dq(mi355x_ctx * ctx, const GemmArgs * members, int n, double bytes, double flops) {
    // two f16 A stages, three B stages, three raw A stages, the Q5_0 bit table; with a GELU epilogue at least the transposition regions + the table's 76 KB
    constexpr uint32_t staging = 2u * BM * 128 + 3u * BN * 128 + 3u * ((dq_raw<AT>::BYTES + 255) & ~255) + 64u;
    constexpr uint32_t with_gelu = (BN / 32) * (2*32*36*4) + GELU_WIN_BYTES;
    static_assert(staging <= 160 * 1024 && with_gelu <= 160 * 1024, "LDS budget of one CU");
    static_assert((BN / 32) * (2*32*36*4) <= 2 * BM * 128 + 3 * BN * 128, "the transposition regions fit the staging area");
    const uint32_t lds = members[0].gelu && with_gelu > staging ? with_gelu : staging;
    const int64_t tiles = (int64_t) ((members[0].M + BM - 1) / BM) * ((members[0].T + BN - 1) / BN);
    const int64_t per = (n * tiles + 7) / 8;
    // (the attribute is set to the ceiling once per instantiation and device: `lds` depends on the epilogue)
    if (n == 1) {
        GemmArgs k = members[0];
        ring_tiling<BM, BN>(k, per);
        k.per = (int) per;
        k.ablate = mi355x_opt(MI355X_OPT_DQ_ABLATE, 0);
        k.dbg = (unsigned long long *) mi355x_debug_stamps(ctx);
        k.lds_bytes = (int) lds;
        static std::atomic<bool> attr_set[64];
        if (ring_lds_attr(ctx, k_gemm_dq<AT, BN>, 160 * 1024, attr_set) != 0) return MI355X_E_UNSUPPORTED;
        return emit(ctx, "gemm_dq", k_gemm_dq<AT, BN>, dim3((uint32_t) (8 * per)), dim3(BN * 2), lds, k, bytes, flops);
    }
    GemmGroupArgs ga; memset(&ga, 0, sizeof(ga));
    for (int i = 0; i < n; i++) { ga.g[i] = members[i]; ring_tiling<BM, BN>(ga.g[i], per); ga.g[i].lds_bytes = (int) lds; }
    ga.n = n; ga.tiles_per_member = (int) tiles; ga.per = (int) per;
    static std::atomic<bool> attr_set_g[64];
    if (ring_lds_attr(ctx, k_gemm_dq_group<AT, BN>, 160 * 1024, attr_set_g) != 0) return MI355X_E_UNSUPPORTED;
    return emit(ctx, "gemm_dq_group", k_gemm_dq_group<AT, BN>, dim3((uint32_t) (8 * per)), dim3(BN * 2), lds, ga, bytes, flops);
}

template <int AT>
static int launch_dq_any(mi355x_ctx * ctx, const GemmArgs * members, int n, double bytes, double flops) {
    return dq_pick_bn(ctx, members[0], n) == 256 ? launch_dq<AT, 256>(ctx, members, n, bytes, flops) : launch_dq<AT, 128>(ctx, members, n, bytes, flops);
}

static bool dq_eligible(const GemmArgs & k, int at) {
    if (at != MI355X_TYPE_Q4_0 && at != MI355X_TYPE_Q5_0 && at != MI355X_TYPE_Q8_0 && at != MI355X_TYPE_Q4_K) return false;
    return k.K % BK == 0 && k.K >= 2*BK && (at != MI355X_TYPE_Q4_K || k.K % 256 == 0) && (k.ldb % 8) == 0 && ((uintptr_t) k.B % 16) == 0 && k.T >= 64 && k.M >= 64 &&
           (k.T + 127) / 128 <= 65535;
}

#else
template <int AT> static int launch_dq_any(mi355x_ctx *, const GemmArgs *, int, double, double) { return MI355X_E_UNSUPPORTED; }
static bool dq_eligible(const GemmArgs &, int) { return false; }
#endif

// ---- held-back ring GEMMs (mi355x_ctx::pending_*) ---------------------------------------------------------------------------
struct PendingGemms { GemmArgs k[GEMM_GROUP_MAX]; double bytes, flops; int at; };     // at: MI355X_TYPE_F16 (ring) or the quantized type (k_gemm_dq)
static_assert(sizeof(PendingGemms) <= sizeof(((mi355x_ctx *) nullptr)->pending_store), "pending_store too small");

static bool mem_overlap(const void * a, int64_t na, const void * b, int64_t nb) {
    return a && b && na > 0 && nb > 0 && (const char *) a < (const char *) b + nb && (const char *) b < (const char *) a + na;
}
static int64_t gemm_dst_bytes(const GemmArgs & k) { return (k.T - 1) * k.dst_nb1 + (int64_t) k.M * (k.dst_f16 ? 2 : 4); }
static int64_t gemm_res_bytes(const GemmArgs & k) { return k.residual ? (k.T - 1) * k.res_nb1 + (int64_t) k.M * 4 : 0; }

// may `k` join the held-back members?  Same B and shape, and no member reads or writes what another member writes.
static bool gemm_mergeable(const PendingGemms & P, int n, const GemmArgs & k, int at) {
    const GemmArgs & f = P.k[0];
    if (k.prep || f.prep || at != P.at) return false;                                   // a product that also writes prepared activations goes alone
    if (n >= GEMM_GROUP_MAX || k.B != f.B || k.ldb != f.ldb || k.K != f.K || k.T != f.T || k.M != f.M) return false;
    const int64_t kd = gemm_dst_bytes(k);
    if (mem_overlap(k.dst, kd, k.B, k.T * k.ldb * 2)) return false;
    for (int i = 0; i < n; i++) {
        const GemmArgs & e = P.k[i];
        const int64_t ed = gemm_dst_bytes(e);
        if (mem_overlap(k.dst, kd, e.dst, ed)) return false;
        if (mem_overlap(k.residual, gemm_res_bytes(k), e.dst, ed) || mem_overlap(e.residual, gemm_res_bytes(e), k.dst, kd)) return false;
        if (mem_overlap(k.A, (int64_t) k.M * k.a_nb1, e.dst, ed) || mem_overlap(e.A, (int64_t) e.M * e.a_nb1, k.dst, kd)) return false;
        if (mem_overlap(k.bias, (k.bias_t ? k.T : (int64_t) k.M) * 4, e.dst, ed) || mem_overlap(e.bias, (e.bias_t ? e.T : (int64_t) e.M) * 4, k.dst, kd)) return false;
    }
    return true;
}

// register-staged MFMA kernel for plain f16 operands (no LDS-DMA ring, static LDS only)
static int launch_staged_f16(mi355x_ctx * ctx, const GemmArgs & k, double bytes, double flops) {
    const int64_t mt = (k.M + BM - 1) / BM, nt128 = (k.T + 127) / 128, nt64 = (k.T + 63) / 64;
    if (mt * nt128 >= ctx->n_cu || k.T > 64*65535LL) {
        if (nt128 > 65535) return MI355X_E_UNSUPPORTED;
        return emit(ctx, "gemm_mfma", k_gemm_mfma<MI355X_TYPE_F16, 128>, dim3((uint32_t) mt, (uint32_t) nt128), dim3(256), 0, k, bytes, flops);
    }
    return emit(ctx, "gemm_mfma", k_gemm_mfma<MI355X_TYPE_F16, 64>, dim3((uint32_t) mt, (uint32_t) nt64), dim3(256), 0, k, bytes, flops);
}

static int flush_pending_gemms(mi355x_ctx * ctx) {
    PendingGemms P; memcpy(&P, ctx->pending_store, sizeof(P));
    const int n = ctx->pending_n;
    ctx->pending_n = 0;
    ctx->in_flush = true;
    const GemmArgs & k = P.k[0];
    const int64_t mt = (k.M + BM - 1) / BM, nt128 = (k.T + 127) / 128;
    int rc;
    if (P.at != MI355X_TYPE_F16) {
        switch (P.at) {
            case MI355X_TYPE_Q4_0: rc = launch_dq_any<MI355X_TYPE_Q4_0>(ctx, P.k, n, P.bytes, P.flops); break;
            case MI355X_TYPE_Q5_0: rc = launch_dq_any<MI355X_TYPE_Q5_0>(ctx, P.k, n, P.bytes, P.flops); break;
            case MI355X_TYPE_Q8_0: rc = launch_dq_any<MI355X_TYPE_Q8_0>(ctx, P.k, n, P.bytes, P.flops); break;
            default:               rc = launch_dq_any<MI355X_TYPE_Q4_K>(ctx, P.k, n, P.bytes, P.flops); break;
        }
        ctx->in_flush = false;
        return rc;
    }
    if (n == 1) {
        // stages: measured on large-v3 encode — 64-wide tiles (one block per CU): 2 -> 13.0 ms, 3 -> 11.3, 4 -> 10.9, 5/6 no better;
        // 128-wide tiles (FC1, ~2 blocks per CU): 2 stages (64 KB, two blocks co-resident) 10.6-10.8 vs 3 -> 10.9, 4 -> 11.0
        // (256-row tiles with 8 waves, and other widths / depths for the chip-covering products, were measured in rounds 2 and 3 and lost: HISTORY.md)
        if (mt * nt128 >= ctx->n_cu) rc = launch_ring<128, 2>(ctx, k, P.bytes, P.flops);
        else                         rc = launch_ring<64, 4>(ctx, k, P.bytes, P.flops);
    } else {
        // measured on large-v3 encode (32 Q/K/V groups + 8 cross-K/V groups of 8, profiles/r02_encoder_ab.txt): 64-wide tiles
        // with a 2-stage ring (48 KB: three workgroups co-resident per CU, each other's DMA waits hidden) 2.13 ms;
        // 64 x 3 stages (2 per CU) 2.81; 128 x 2 stages (2 per CU) 2.63; 64 x 4 (1 per CU) 3.51; 128 x 3 (1 per CU) 3.88;
        // the same products as single launches (64 x 4) 2.9
        rc = launch_ring_group<64, 2>(ctx, P.k, n, P.bytes, P.flops);
    }
    if (rc == MI355X_E_UNSUPPORTED) {
        // the runtime rejected the ring kernel's dynamic LDS size: every member leaves as the register-staged kernel instead (the
        // path launch_gemm takes when the ring is switched off) — same tiles, same results, nothing surfaces on a later op
        rc = 0;
        for (int i = 0; i < n && rc == 0; i++) rc = launch_staged_f16(ctx, P.k[i], P.bytes / n, P.flops / n);
    }
    ctx->in_flush = false;
    return rc;
}

// hold a ring GEMM back; it leaves with the next flush (mi355x_flush_pending: any other launch, synchronize, end of the graph range)
static int hold_ring_gemm(mi355x_ctx * ctx, const GemmArgs & k, double bytes, double flops, int at = MI355X_TYPE_F16) {
    PendingGemms * P = (PendingGemms *) ctx->pending_store;
    // (the pending store may hold another kernel family's launches — mmq.hip groups its products the same way: those leave first)
    if (ctx->pending_n > 0 && (ctx->pending_flush != flush_pending_gemms || !gemm_mergeable(*P, ctx->pending_n, k, at))) {
        const int rc = mi355x_flush_pending(ctx);
        if (rc) return rc;
    }
    if (ctx->pending_n == 0) { P->bytes = 0; P->flops = 0; P->at = at; }
    P->k[ctx->pending_n++] = k;
    P->bytes += bytes; P->flops += flops;
    ctx->pending_flush = flush_pending_gemms;
    return 0;
}

template <int AT>
static int launch_gemm(mi355x_ctx * ctx, const GemmArgs & k, double bytes, double flops) {
    // pick BN so the grid covers the chip: 128-wide token tiles unless that leaves CUs idle
    const int64_t mt = (k.M + BM - 1) / BM;
    const int64_t nt128 = (k.T + 127) / 128, nt64 = (k.T + 63) / 64;
    if constexpr (AT == MI355X_TYPE_F16) {
        // both operands plain f16: the LDS-DMA ring
        if (k.K % BK == 0 && k.K >= 2*BK && (k.a_nb1 % 16) == 0 && ((uintptr_t) k.A % 16) == 0 && (k.ldb % 8) == 0 && nt64 <= 65535) {
            return hold_ring_gemm(ctx, k, bytes, flops);
        }
    } else if (dq_eligible(k, AT) && mi355x_opt(MI355X_OPT_DQ_GEMM, 0) != 0) {
        // quantized weight x wide f16 activations: dequantized per workgroup on its way into LDS (k_gemm_dq)
        return hold_ring_gemm(ctx, k, bytes, flops, AT);
    }
    if (mt * nt128 >= ctx->n_cu || k.T > 64*65535LL) {
        if (nt128 > 65535) return MI355X_E_UNSUPPORTED;
        return emit(ctx, "gemm_mfma", k_gemm_mfma<AT, 128>, dim3((uint32_t) mt, (uint32_t) nt128), dim3(256), 0, k, bytes, flops);
    }
    return emit(ctx, "gemm_mfma", k_gemm_mfma<AT, 64>, dim3((uint32_t) mt, (uint32_t) nt64), dim3(256), 0, k, bytes, flops);
}

// A: [K, M] ggml src0; Bf16: prepared [T][K]; dst column stride dst_nb1
static int gemm_f16act_impl(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act, int64_t ldb, int64_t T,
                            void * dst, int64_t dst_nb1, int dst_type, const mi355x_epilogue * ep, void * prep_out, int prep_only);

extern "C" int mi355x_gemm_f16act(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act, int64_t ldb, int64_t T,
                                  void * dst, int64_t dst_nb1, int dst_type, const mi355x_epilogue * ep) {
    return gemm_f16act_impl(ctx, A, act, ldb, T, dst, dst_nb1, dst_type, ep, nullptr, 0);
}

// the same product, whose epilogue ALSO leaves mi355x_prep_act(mode 1) of the F32 result in prep_out (f16 [T][M]): the result is the
// activation matrix of the next GEMM.  dst may be NULL when nothing else reads the F32 result (then only prep_out is written).
extern "C" int mi355x_gemm_f16act_prep(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act, int64_t ldb, int64_t T,
                                       void * dst, int64_t dst_nb1, const mi355x_epilogue * ep, void * prep_out) {
    const int64_t M = A->ne[1];
    if (!prep_out || ((uintptr_t) prep_out % 16) || M % 32) return MI355X_E_UNSUPPORTED;
    if (dst && (((uintptr_t) dst % 16) || (dst_nb1 % 16))) return MI355X_E_UNSUPPORTED;
    if (ep && ((ep->bias && !ep->bias_per_col && ((uintptr_t) ep->bias % 16)) || (ep->residual && (((uintptr_t) ep->residual % 16) || (ep->residual_nb1 % 16))))) return MI355X_E_UNSUPPORTED;
    return gemm_f16act_impl(ctx, A, act, ldb, T, dst ? dst : prep_out, dst ? dst_nb1 : M*4, MI355X_TYPE_F32, ep, prep_out, dst ? 0 : 1);
}

static int gemm_f16act_impl(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act, int64_t ldb, int64_t T,
                            void * dst, int64_t dst_nb1, int dst_type, const mi355x_epilogue * ep, void * prep_out, int prep_only) {
    if (dst_type != MI355X_TYPE_F32 && dst_type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
    const uint16_t * Bf16 = (const uint16_t *) act; const int dst_f16 = dst_type == MI355X_TYPE_F16;
    GemmArgs k; memset(&k, 0, sizeof(k));
    k.prep = (uint16_t *) prep_out; k.prep_only = prep_only;
    const int K = (int) A->ne[0], M = (int) A->ne[1];
    k.A = (const char *) A->data; k.a_nb1 = A->nb[1]; k.B = Bf16; k.ldb = ldb; k.M = M; k.K = K; k.T = T;
    k.dst = (char *) dst; k.dst_nb1 = dst_nb1; k.dst_f16 = dst_f16; k.gelu_tab = ctx->gelu_tab;
    if (ep) { k.bias_t = ep->bias && ep->bias_per_col; k.bias = ep->bias; k.scale = ep->scale; k.has_scale = ep->has_scale; k.gelu = ep->gelu; k.residual = (const char *) ep->residual; k.res_nb1 = ep->residual_nb1; }
    if (M <= 0 || T <= 0 || K <= 0 || K % 8 || ldb % 8 || ((uintptr_t) Bf16 % 16)) return MI355X_E_UNSUPPORTED;
    const double flops = 2.0 * M * (double) K * (double) T;
    double abytes;
    if (mi355x_type_is_quantized(A->type)) {
        if (!t_is_contiguous(A) || K % type_block(A->type) || ((uintptr_t) A->data % 16)) return MI355X_E_UNSUPPORTED;
        k.nbt = (int64_t) M * (K / type_block(A->type));
        abytes = (double) mi355x_type_row_bytes(A->type, K) * M;
    } else {
        const int es = A->type == MI355X_TYPE_F16 ? 2 : 4;
        if ((A->type != MI355X_TYPE_F16 && A->type != MI355X_TYPE_F32) || A->nb[0] != es || (A->nb[1] % 16) || ((uintptr_t) A->data % 16)) return MI355X_E_UNSUPPORTED;
        abytes = (double) M * K * es;
    }
    const double bytes = abytes + (double) T*K*2 + (double) T*M*(dst_f16 ? 2 : 4);
    switch (A->type) {
        case MI355X_TYPE_Q4_0: return launch_gemm<MI355X_TYPE_Q4_0>(ctx, k, bytes, flops);
        case MI355X_TYPE_Q5_0: return launch_gemm<MI355X_TYPE_Q5_0>(ctx, k, bytes, flops);
        case MI355X_TYPE_Q8_0: return launch_gemm<MI355X_TYPE_Q8_0>(ctx, k, bytes, flops);
        case MI355X_TYPE_Q4_K: return launch_gemm<MI355X_TYPE_Q4_K>(ctx, k, bytes, flops);
        case MI355X_TYPE_F16:  return launch_gemm<MI355X_TYPE_F16>(ctx, k, bytes, flops);
        case MI355X_TYPE_F32:  return launch_gemm<MI355X_TYPE_F32>(ctx, k, bytes, flops);
        default: return MI355X_E_UNSUPPORTED;
    }
}
