// Internal header of libmi355x_kernels.so: context, launch emission, device helpers.
// gfx950 only (wave64, MFMA, v_dot4_i32_i8).  No CUDA compatibility layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>

#include "mi355x_kernels.h"

#define WAVE 64

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float    floatx4  __attribute__((ext_vector_type(4)));
typedef float    floatx16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct prof_acc { uint64_t calls = 0; double ms = 0, bytes = 0, flops = 0; };

struct mi355x_ctx {
    int         device   = 0;
    hipStream_t stream   = nullptr;
    // constant tables
    uint16_t *  gelu_tab = nullptr;      // 65536 x f16 (device)
    // scratch arena (grown on demand; only touched from this ctx's stream, so reuse is stream-ordered)
    void *      scratch      = nullptr;
    size_t      scratch_size = 0;
    size_t      scratch_used = 0;        // bump pointer, reset per op group
    std::vector<void *> scratch_retired; // outgrown arenas that may still back pointers of the current op group
    int         last_mirrored = 0;      // the last mat-vec launch also wrote its caller's mirror columns (mi355x_last_launch_mirrored)
    void *      qact = nullptr;          // two quantized-activation plane buffers (decode_q.hip, mi355x_act_scratch)
    float *     mel_tab = nullptr;       // sin / cos / Hann tables + the running maximum of mi355x_log_mel (device)
    uint64_t    n_eager = 0;             // launches issued directly on the stream so far (mi355x_eager_count)
    // profiling
    bool                                 prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    struct pending { const char * name; int ev; double bytes, flops; };
    std::vector<pending>                 ev_pending;
    std::map<std::string, prof_acc>      prof_rows;
    int                                  n_cu = 256;
    void *                               dbg_stamps = nullptr;   // 16 x u64 (device), GGML_MI355X_KTIME=1 only
    // launches held back so that independent neighbours can go out as ONE grouped launch (gemm_mfma.hip: the Q / K / V projections
    // of an encoder layer, the cross-attention K / V projections of consecutive layers).  Anything else that is emitted, and every
    // operation that orders against the stream, flushes them first (mi355x_flush_pending), so stream order is unchanged.
    int      pending_n = 0;
    bool     in_flush  = false;
    int   (* pending_flush)(mi355x_ctx *) = nullptr;
    alignas(16) uint8_t pending_store[2048];
};
static inline int mi355x_flush_pending(mi355x_ctx * ctx) { return (ctx->pending_n > 0 && !ctx->in_flush && ctx->pending_flush) ? ctx->pending_flush(ctx) : 0; }
void * mi355x_debug_stamps(mi355x_ctx * ctx);
int    mi355x_opt(int opt, int def);          // value of a test option (mi355x_test_option), `def` when unset

void   mi355x_set_error(const char * fmt, ...);
#define HIP_CHECK_RET(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { mi355x_set_error("%s failed: %s", #call, hipGetErrorString(e_)); return (int) e_; } } while (0)
// scratch: returns a device pointer valid until the NEXT mi355x_scratch_reset on this ctx
void * mi355x_scratch_alloc(mi355x_ctx * ctx, size_t bytes);
void   mi355x_scratch_reset(mi355x_ctx * ctx);

// emit one kernel launch (hipLaunchKernel on ctx->stream)
int mi355x_emit(mi355x_ctx * ctx, const char * name, const void * func, dim3 grid, dim3 block, uint32_t shmem,
                const void * args, uint32_t arg_size, double algo_bytes, double algo_flops);

template <typename Args>
static inline int emit(mi355x_ctx * ctx, const char * name, void (*kernel)(Args), dim3 grid, dim3 block, uint32_t shmem,
                       const Args & a, double bytes = 0, double flops = 0) {
    return mi355x_emit(ctx, name, (const void *) kernel, grid, block, shmem, &a, (uint32_t) sizeof(Args), bytes, flops);
}

// second-generation decoder mat-vec (decode.hip); MI355X_E_UNSUPPORTED => caller uses k_gemv (gemv.hip)
int mi355x_gemv8(mi355x_ctx * ctx, const mi355x_gemv_desc * d);
// mat-vec over pre-quantized activation planes (decode_q.hip); MI355X_E_UNSUPPORTED => mi355x_gemv8 copies the same planes (k_gemv8)
int mi355x_gemv_q(mi355x_ctx * ctx, const mi355x_gemv_desc * d);
int mi355x_vocab(mi355x_ctx * ctx, const mi355x_gemv_desc * d);       // decode_q.hip: the vocabulary projection (N > 8192)
// mat-vec over prepared planes on the matrix cores, 9..32 columns (decode_mx.hip); MI355X_E_UNSUPPORTED => mi355x_vocab / mi355x_gemv_q
int mi355x_gemv_mx(mi355x_ctx * ctx, const mi355x_gemv_desc * d);

// ---------------------------------------------------------------------------------------------
// tensor helpers (host)
// ---------------------------------------------------------------------------------------------
static inline int64_t t_nelements(const mi355x_tensor * t) { return t->ne[0]*t->ne[1]*t->ne[2]*t->ne[3]; }
static inline int64_t t_nrows(const mi355x_tensor * t)     { return t->ne[1]*t->ne[2]*t->ne[3]; }
static inline int     type_block(int type) {
    switch (type) { case MI355X_TYPE_Q4_0: case MI355X_TYPE_Q5_0: case MI355X_TYPE_Q8_0: return 32; case MI355X_TYPE_Q4_K: return 256; default: return 1; }
}
static inline int     type_size(int type) {   // bytes per block
    switch (type) {
        case MI355X_TYPE_F32: case MI355X_TYPE_I32: return 4;
        case MI355X_TYPE_F16: return 2;
        case MI355X_TYPE_Q4_0: return 18; case MI355X_TYPE_Q5_0: return 22; case MI355X_TYPE_Q8_0: return 34; case MI355X_TYPE_Q4_K: return 144;
        default: return 0;
    }
}
static inline bool t_is_contiguous(const mi355x_tensor * t) {
    int64_t nb = type_size(t->type);
    if (t->nb[0] != nb) return false;
    nb = nb * (t->ne[0] / type_block(t->type));
    for (int i = 1; i < 4; i++) { if (t->ne[i] != 1 && t->nb[i] != nb) return false; nb *= t->ne[i]; }
    return true;
}
static inline bool t_same_shape(const mi355x_tensor * a, const mi355x_tensor * b) {
    return a->ne[0]==b->ne[0] && a->ne[1]==b->ne[1] && a->ne[2]==b->ne[2] && a->ne[3]==b->ne[3];
}

// device-side compact tensor view (by value in kernel args)
struct dtensor {
    char *  data;
    int64_t ne[4];
    int64_t nb[4];
};
static inline dtensor to_d(const mi355x_tensor * t) {
    dtensor d; d.data = (char *) t->data;
    for (int i = 0; i < 4; i++) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    return d;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ float h2f(uint16_t h) { half_t x; __builtin_memcpy(&x, &h, 2); return (float) x; }
__device__ __forceinline__ uint16_t f2h(float f) { half_t x = (half_t) f; uint16_t h; __builtin_memcpy(&h, &x, 2); return h; }
__device__ __forceinline__ float round_f16(float f) { return (float) (half_t) f; }

// ---- cross-lane exchange on the VALU (no LDS crossbar): DPP inside a 16-lane row, v_permlane{16,32}_swap across rows.
// dpp_x<CTRL>(v) returns v of the partner lane: 0xB1 = quad_perm[1,0,3,2] (lane^1), 0x4E = quad_perm[2,3,0,1] (lane^2),
// 0x141 = row_half_mirror (i <-> 7-i), 0x140 = row_mirror (i <-> 15-i).  In a butterfly reduction the mirror partners
// hold the same partial result as the xor-4 / xor-8 partners would, so the sums are bit-identical to a __shfl_xor
// butterfly at a fraction of its latency (a ds_bpermute round trip per step).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp_x(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_xi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// value of lane^16 / lane^32
__device__ __forceinline__ float lane_xor16(float v) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0] ^ r[1] ^ __float_as_uint(v));       // {own, partner} in some order: xor out the own value
}
__device__ __forceinline__ float lane_xor32(float v) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0] ^ r[1] ^ __float_as_uint(v));
}
__device__ __forceinline__ int lane_xor16_i(int v) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap((unsigned) v, (unsigned) v, false, false);
    return (int) (r[0] ^ r[1] ^ (unsigned) v);
}
__device__ __forceinline__ int lane_xor32_i(int v) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap((unsigned) v, (unsigned) v, false, false);
    return (int) (r[0] ^ r[1] ^ (unsigned) v);
}
// reductions over aligned groups of G lanes (G = 2..64), result in every lane of the group
template <int G> __device__ __forceinline__ float group_sum(float v) {
    if (G >= 2)  v += dpp_x<0xB1>(v);
    if (G >= 4)  v += dpp_x<0x4E>(v);
    if (G >= 8)  v += dpp_x<0x141>(v);
    if (G >= 16) v += dpp_x<0x140>(v);
    // after the swap the two results are {own row's value, partner row's value} in some order: add both
    if (G >= 32) { const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    if (G >= 64) { const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
template <int G> __device__ __forceinline__ float group_max(float v) {
    if (G >= 2)  v = fmaxf(v, dpp_x<0xB1>(v));
    if (G >= 4)  v = fmaxf(v, dpp_x<0x4E>(v));
    if (G >= 8)  v = fmaxf(v, dpp_x<0x141>(v));
    if (G >= 16) v = fmaxf(v, dpp_x<0x140>(v));
    if (G >= 32) { const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    if (G >= 64) { const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    return v;
}
template <int G> __device__ __forceinline__ int group_sum_i(int v) {
    if (G >= 2)  v += dpp_xi<0xB1>(v);
    if (G >= 4)  v += dpp_xi<0x4E>(v);
    if (G >= 8)  v += dpp_xi<0x141>(v);
    if (G >= 16) v += dpp_xi<0x140>(v);
    if (G >= 32) { const u32x2_t r = __builtin_amdgcn_permlane16_swap((unsigned) v, (unsigned) v, false, false); v = (int) (r[0] + r[1]); }
    if (G >= 64) { const u32x2_t r = __builtin_amdgcn_permlane32_swap((unsigned) v, (unsigned) v, false, false); v = (int) (r[0] + r[1]); }
    return v;
}
// sum / max over the 8 lanes that share (lane & 7): lane^8 (row_ror:8), lane^16, lane^32
__device__ __forceinline__ float stride8_sum(float v) {
    v += dpp_x<0x128>(v);
    { const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
__device__ __forceinline__ float stride8_max(float v) {
    v = fmaxf(v, dpp_x<0x128>(v));
    { const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    { const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }
__device__ __forceinline__ int wave_sum_i(int v) { return group_sum_i<64>(v); }

// spread the low 4 bits of b to bit 4 of each byte of a dword: bit k -> byte k, bit 4
__device__ __forceinline__ uint32_t spread4_to_bit4(uint32_t b) {
    return (((b & 0xFu) * 0x00204081u) & 0x01010101u) << 4;
}

// f16-table GELU exactly as ggml_vec_gelu_f32 (ggml-cpu/vec.h:987-1000)
__device__ __forceinline__ float gelu_lut(float x, const uint16_t * __restrict__ tab) {
    if (x <= -10.0f) return 0.0f;
    if (x >=  10.0f) return x;
    return h2f(tab[f2h(x)]);
}

// ---- planar quantized layouts (see mi355x_kernels.h) -------------------------------------------
// nbt = total number of blocks in the tensor
template <int TYPE> struct qplanes;
template <> struct qplanes<MI355X_TYPE_Q4_0> {
    const uint8_t * qs; const uint16_t * d;
    __device__ __forceinline__ qplanes(const void * base, int64_t nbt) { qs = (const uint8_t *) base; d = (const uint16_t *) (qs + nbt*16); }
};
template <> struct qplanes<MI355X_TYPE_Q5_0> {
    const uint8_t * qs; const uint32_t * qh; const uint16_t * d;
    __device__ __forceinline__ qplanes(const void * base, int64_t nbt) { qs = (const uint8_t *) base; qh = (const uint32_t *) (qs + nbt*16); d = (const uint16_t *) (qs + nbt*20); }
};
template <> struct qplanes<MI355X_TYPE_Q8_0> {
    const uint8_t * qs; const uint16_t * d;
    __device__ __forceinline__ qplanes(const void * base, int64_t nbt) { qs = (const uint8_t *) base; d = (const uint16_t *) (qs + nbt*32); }
};
template <> struct qplanes<MI355X_TYPE_Q4_K> {
    const uint8_t * qs; const uint8_t * sc; const uint32_t * dm;
    __device__ __forceinline__ qplanes(const void * base, int64_t nbt) { qs = (const uint8_t *) base; sc = qs + nbt*128; dm = (const uint32_t *) (qs + nbt*140); }
};

// Q4_K 6-bit scale/min extraction (get_scale_min_k4, ggml/src/ggml-quants.c:880-887); q = 12 scale bytes
__device__ __forceinline__ void q4k_scale_min(int j, const uint8_t * q, int & sc, int & m) {
    if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
    else       { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

// the same from the 12 scale bytes held as three little-endian words (registers): shifts and selects only — indexing a
// register-resident byte array with a lane-dependent j would put it in scratch memory
__device__ __forceinline__ void q4k_scale_min_w(int j, uint32_t s0, uint32_t s1, uint32_t s2, int & sc, int & m) {
    const int sh = (j & 3) * 8;
    const int a_sc = (int) ((s0 >> sh) & 63), a_m = (int) ((s1 >> sh) & 63);
    const int b_sc = (int) (((s2 >> sh) & 0xF) | (((s0 >> (sh + 6)) & 3) << 4));
    const int b_m  = (int) (((s2 >> (sh + 4)) & 0xF) | (((s1 >> (sh + 6)) & 3) << 4));
    sc = j < 4 ? a_sc : b_sc;
    m  = j < 4 ? a_m  : b_m;
}

// dequantize element-block helpers: write 32 floats of block `ib` (32-element granularity for all types;
// for Q4_K `ib` indexes 32-element sub-blocks: super-block ib/8, sub-block ib%8)
template <int TYPE>
__device__ __forceinline__ void dequant_block32(const qplanes<TYPE> & p, int64_t ib, float * out);

template <>
__device__ __forceinline__ void dequant_block32<MI355X_TYPE_Q4_0>(const qplanes<MI355X_TYPE_Q4_0> & p, int64_t ib, float * out) {
    const uint4 q = *(const uint4 *) (p.qs + ib*16);
    const float d = h2f(p.d[ib]);
    const uint32_t w[4] = { q.x, q.y, q.z, q.w };
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        #pragma unroll
        for (int b = 0; b < 4; b++) {
            const int byte = (w[i] >> (8*b)) & 0xFF;
            out[4*i + b]      = (float) ((byte & 0xF) - 8) * d;     // dequantize_row_q4_0, ggml-quants.c:459-477
            out[4*i + b + 16] = (float) ((byte >> 4)  - 8) * d;
        }
    }
}
template <>
__device__ __forceinline__ void dequant_block32<MI355X_TYPE_Q5_0>(const qplanes<MI355X_TYPE_Q5_0> & p, int64_t ib, float * out) {
    const uint4 q = *(const uint4 *) (p.qs + ib*16);
    const uint32_t qh = p.qh[ib];
    const float d = h2f(p.d[ib]);
    const uint32_t w[4] = { q.x, q.y, q.z, q.w };
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        #pragma unroll
        for (int b = 0; b < 4; b++) {
            const int j = 4*i + b;
            const int byte = (w[i] >> (8*b)) & 0xFF;
            const int h0 = ((qh >> j) & 1) << 4;                     // dequantize_row_q5_0, ggml-quants.c:500-524
            const int h1 = ((qh >> (j + 16)) & 1) << 4;
            out[j]      = (float) (((byte & 0xF) | h0) - 16) * d;
            out[j + 16] = (float) (((byte >> 4)  | h1) - 16) * d;
        }
    }
}
template <>
__device__ __forceinline__ void dequant_block32<MI355X_TYPE_Q8_0>(const qplanes<MI355X_TYPE_Q8_0> & p, int64_t ib, float * out) {
    const uint4 q0 = *(const uint4 *) (p.qs + ib*32);
    const uint4 q1 = *(const uint4 *) (p.qs + ib*32 + 16);
    const float d = h2f(p.d[ib]);
    const uint32_t w[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
    #pragma unroll
    for (int i = 0; i < 8; i++) {
        #pragma unroll
        for (int b = 0; b < 4; b++) out[4*i + b] = (float) (int8_t) ((w[i] >> (8*b)) & 0xFF) * d;   // ggml-quants.c:553-567
    }
}
template <>
__device__ __forceinline__ void dequant_block32<MI355X_TYPE_Q4_K>(const qplanes<MI355X_TYPE_Q4_K> & p, int64_t ib, float * out) {
    const int64_t sb = ib >> 3; const int j = (int) (ib & 7);
    const uint32_t dm = p.dm[sb];
    const float d = h2f((uint16_t) (dm & 0xFFFF)), dmin = h2f((uint16_t) (dm >> 16));
    int sc, m; q4k_scale_min(j, p.sc + sb*12, sc, m);
    const float d1 = d * sc, m1 = dmin * m;                          // dequantize_row_q4_K, ggml-quants.c:1529-1551
    const uint8_t * qs = p.qs + sb*128 + (j >> 1)*32;
    const uint4 q0 = *(const uint4 *) qs, q1 = *(const uint4 *) (qs + 16);
    const uint32_t w[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
    const int sh = (j & 1) * 4;
    #pragma unroll
    for (int i = 0; i < 8; i++) {
        #pragma unroll
        for (int b = 0; b < 4; b++) out[4*i + b] = d1 * (float) ((w[i] >> (8*b + sh)) & 0xF) - m1;
    }
}

#endif // __HIPCC__
