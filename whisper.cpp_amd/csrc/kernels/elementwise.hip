// Bandwidth-class ops of the whisper graphs: add/mul (broadcast), scale, gelu, norm(+affine), cpy/cont/cast,
// get_rows, im2col(1-D), soft_max, rope, concat.  All HBM/L2-bound: coalesced 16-byte accesses on the
// contiguous fast paths, wave64 reductions, no LDS except block reductions.
#include "common.h"
#include "qrows.h"
#include <math.h>

// -------------------------------------------------------------------------------------------------
// binary add/mul with ggml broadcasting of src1 (ggml-cpu/binary-ops.cpp:140-148)
// -------------------------------------------------------------------------------------------------
struct BinArgs { dtensor a, b, d; int op; int64_t nchunk0; };

template <int OP> __device__ __forceinline__ float bin_op(float x, float y) { return OP == 0 ? x + y : x * y; }

// generic: one thread per element of a row chunk; rows flattened into blockIdx.x
__global__ void __launch_bounds__(256) k_bin_generic(const BinArgs a) {
    const int64_t row = blockIdx.x / a.nchunk0;
    const int64_t i0  = (blockIdx.x % a.nchunk0) * 256 + threadIdx.x;
    if (i0 >= a.d.ne[0]) return;
    const int64_t i1 = row % a.d.ne[1], i2 = (row / a.d.ne[1]) % a.d.ne[2], i3 = row / (a.d.ne[1]*a.d.ne[2]);
    const float x = *(const float *) (a.a.data + i0*a.a.nb[0] + i1*a.a.nb[1] + i2*a.a.nb[2] + i3*a.a.nb[3]);
    const float y = *(const float *) (a.b.data + (i0 % a.b.ne[0])*a.b.nb[0] + (i1 % a.b.ne[1])*a.b.nb[1] + (i2 % a.b.ne[2])*a.b.nb[2] + (i3 % a.b.ne[3])*a.b.nb[3]);
    float * o = (float *) (a.d.data + i0*a.d.nb[0] + i1*a.d.nb[1] + i2*a.d.nb[2] + i3*a.d.nb[3]);
    *o = a.op == 0 ? x + y : x * y;
}

// fast path: a, d contiguous f32 with ne0 % 4 == 0; b either same-shape contiguous (bmode 0) or a
// contiguous row vector of ne0 broadcast over all rows (bmode 1)
struct BinFastArgs { const float * a; const float * b; float * d; int64_t n4; int64_t ne0_4; int op; int bmode; };
__global__ void __launch_bounds__(256) k_bin_fast(const BinFastArgs a) {
    const float4 * pa = (const float4 *) a.a; const float4 * pb = (const float4 *) a.b; float4 * pd = (float4 *) a.d;
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (int64_t) gridDim.x * 256) {
        const float4 x = pa[i];
        const float4 y = a.bmode == 0 ? pb[i] : pb[i % a.ne0_4];
        float4 r;
        if (a.op == 0) { r.x = x.x + y.x; r.y = x.y + y.y; r.z = x.z + y.z; r.w = x.w + y.w; }
        else           { r.x = x.x * y.x; r.y = x.y * y.y; r.z = x.z * y.z; r.w = x.w * y.w; }
        pd[i] = r;
    }
}

extern "C" int mi355x_binary(mi355x_ctx * ctx, int op, const mi355x_tensor * a, const mi355x_tensor * b, const mi355x_tensor * d) {
    if (a->type != MI355X_TYPE_F32 || b->type != MI355X_TYPE_F32 || d->type != MI355X_TYPE_F32) return MI355X_E_UNSUPPORTED;
    if (!t_same_shape(a, d)) return MI355X_E_UNSUPPORTED;
    for (int i = 0; i < 4; i++) if (b->ne[i] <= 0 || d->ne[i] % b->ne[i]) return MI355X_E_UNSUPPORTED;
    const int64_t n = t_nelements(d);
    if (n == 0) return 0;
    const bool ca = t_is_contiguous(a) && t_is_contiguous(d) && (d->ne[0] % 4 == 0) &&
                    ((uintptr_t) a->data % 16 == 0) && ((uintptr_t) d->data % 16 == 0) && ((uintptr_t) b->data % 16 == 0);
    if (ca && t_is_contiguous(b)) {
        int bmode = -1;
        if (t_same_shape(a, b)) bmode = 0;
        else if (b->ne[0] == d->ne[0] && t_nrows(b) == 1) bmode = 1;
        if (bmode >= 0) {
            BinFastArgs k = { (const float *) a->data, (const float *) b->data, (float *) d->data, n/4, d->ne[0]/4, op, bmode };
            const int64_t nb = (n/4 + 255) / 256;
            const int grid = (int) (nb < 4096 ? nb : 4096);
            return emit(ctx, op == 0 ? "add" : "mul", k_bin_fast, dim3(grid), dim3(256), 0, k, (double) n * (bmode == 0 ? 12 : 8), 0);
        }
    }
    BinArgs k = { to_d(a), to_d(b), to_d(d), op, (d->ne[0] + 255) / 256 };
    const int64_t nblocks = k.nchunk0 * t_nrows(d);
    if (nblocks > 0x7fffffffLL) return MI355X_E_UNSUPPORTED;
    return emit(ctx, op == 0 ? "add" : "mul", k_bin_generic, dim3((uint32_t) nblocks), dim3(256), 0, k, (double) n * 12, 0);
}

// -------------------------------------------------------------------------------------------------
// scale (ggml-cpu/ops.cpp:4568-4620): y = x*s + b   (s*x first, then +b when b != 0 — ggml_vec_mad1_f32)
// gelu  (ggml-cpu/vec.h:987-1000)
// -------------------------------------------------------------------------------------------------
struct UnaryArgs { const float * x; float * y; int64_t n; float s, b; int mode; const uint16_t * tab; };
// mode 0 scale (+ bias), 1 GELU (f16 table), 2 ReLU, 3 sigmoid, 4 tanh, 5 sqrt — ggml-cpu/unary-ops.cpp:19-53 (op_tanh, op_relu, op_sigmoid, op_sqrt)
__device__ __forceinline__ float unary_apply(float x, const UnaryArgs & a) {
    switch (a.mode) {
        case 0:  return a.b == 0.0f ? x*a.s : x*a.s + a.b;
        case 1:  return gelu_lut(x, a.tab);
        case 2:  return x > 0.0f ? x : 0.0f;
        case 3:  return 1.0f / (1.0f + expf(-x));
        case 4:  return tanhf(x);
        default: return sqrtf(x);
    }
}
__global__ void __launch_bounds__(256) k_unary(const UnaryArgs a) {
    for (int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (int64_t) gridDim.x * 1024) {
        float v[4];
        if (i + 4 <= a.n && ((uintptr_t) (a.x + i) % 16 == 0) && ((uintptr_t) (a.y + i) % 16 == 0)) {
            const float4 x = *(const float4 *) (a.x + i);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            #pragma unroll
            for (int j = 0; j < 4; j++) v[j] = unary_apply(v[j], a);
            *(float4 *) (a.y + i) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int j = 0; j < 4 && i + j < a.n; j++) {
                const float x = a.x[i + j];
                a.y[i + j] = unary_apply(x, a);
            }
        }
    }
}
static int unary_launch(mi355x_ctx * ctx, const char * name, const mi355x_tensor * x, const mi355x_tensor * y, int mode, float s, float b) {
    if (x->type != MI355X_TYPE_F32 || y->type != MI355X_TYPE_F32 || !t_is_contiguous(x) || !t_is_contiguous(y) || !t_same_shape(x, y)) return MI355X_E_UNSUPPORTED;
    const int64_t n = t_nelements(x);
    if (n == 0) return 0;
    UnaryArgs k = { (const float *) x->data, (float *) y->data, n, s, b, mode, ctx->gelu_tab };
    const int64_t nb = (n + 1023) / 1024;
    return emit(ctx, name, k_unary, dim3((uint32_t) (nb < 4096 ? nb : 4096)), dim3(256), 0, k, (double) n * 8, 0);
}
extern "C" int mi355x_scale(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * y, float s, float b) { return unary_launch(ctx, "scale", x, y, 0, s, b); }
extern "C" int mi355x_gelu(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * y) { return unary_launch(ctx, "gelu", x, y, 1, 0, 0); }
extern "C" int mi355x_unary(mi355x_ctx * ctx, int op, const mi355x_tensor * x, const mi355x_tensor * y) {
    if (op < MI355X_UNARY_RELU || op > MI355X_UNARY_SQRT) return MI355X_E_UNSUPPORTED;
    return unary_launch(ctx, "unary", x, y, op, 0, 0);
}

// -------------------------------------------------------------------------------------------------
// pad_reflect_1d (ggml-cpu/ops.cpp:8149-8180): dst[p0 + i] = x[i];  dst[p0 - k] = x[k];  dst[p0 + n - 1 + k] = x[n - 1 - k]
// -------------------------------------------------------------------------------------------------
struct PadReflectArgs { dtensor x, y; int p0, p1; int64_t n; };
__global__ void __launch_bounds__(256) k_pad_reflect_1d(const PadReflectArgs a) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t) gridDim.x * 256) {
        int64_t r = i;
        const int64_t i0 = r % a.y.ne[0]; r /= a.y.ne[0];
        const int64_t i1 = r % a.y.ne[1]; r /= a.y.ne[1];
        const int64_t i2 = r % a.y.ne[2], i3 = r / a.y.ne[2];
        const int64_t n0 = a.x.ne[0];
        int64_t j = i0 - a.p0;
        if (j < 0) j = -j; else if (j >= n0) j = 2*(n0 - 1) - j;
        const float v = *(const float *) (a.x.data + j*a.x.nb[0] + i1*a.x.nb[1] + i2*a.x.nb[2] + i3*a.x.nb[3]);
        *(float *) (a.y.data + i0*a.y.nb[0] + i1*a.y.nb[1] + i2*a.y.nb[2] + i3*a.y.nb[3]) = v;
    }
}
extern "C" int mi355x_pad_reflect_1d(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * y, int p0, int p1) {
    if (x->type != MI355X_TYPE_F32 || y->type != MI355X_TYPE_F32 || p0 < 0 || p1 < 0 || p0 >= x->ne[0] || p1 >= x->ne[0] || y->ne[0] != x->ne[0] + p0 + p1 ||
        y->ne[1] != x->ne[1] || y->ne[2] != x->ne[2] || y->ne[3] != x->ne[3]) return MI355X_E_UNSUPPORTED;
    PadReflectArgs k = { to_d(x), to_d(y), p0, p1, t_nelements(y) };
    if (k.n == 0) return 0;
    const int64_t nb = (k.n + 255) / 256;
    return emit(ctx, "pad_reflect_1d", k_pad_reflect_1d, dim3((uint32_t) (nb < 8192 ? nb : 8192)), dim3(256), 0, k, (double) k.n * 8, 0);
}

// -------------------------------------------------------------------------------------------------
// norm (+ optional affine): one wave per row (ggml-cpu/ops.cpp:3698-3765)
//   mean = sum/n ; var = sum((x-mean)^2)/n ; y = (x-mean) * (1/sqrtf(var+eps)) [* w + b as separate roundings]
// -------------------------------------------------------------------------------------------------
struct NormArgs { dtensor x, y; float eps; const float * w; const float * b; int64_t nrows; uint16_t * prep; int prep_mode; int8_t * rq; float * rd; int * rs; };
__global__ void __launch_bounds__(256) k_norm(const NormArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.nrows) return;
    const int64_t i1 = row % a.x.ne[1], i2 = (row / a.x.ne[1]) % a.x.ne[2], i3 = row / (a.x.ne[1]*a.x.ne[2]);
    const float * x = (const float *) (a.x.data + i1*a.x.nb[1] + i2*a.x.nb[2] + i3*a.x.nb[3]);
    float * y = (float *) (a.y.data + i1*a.y.nb[1] + i2*a.y.nb[2] + i3*a.y.nb[3]);
    const int n = (int) a.x.ne[0];
    float s = 0.0f;
    for (int i = lane; i < n; i += 64) s += x[i];
    s = wave_sum(s);
    const float mean = s / n;
    float v = 0.0f;
    for (int i = lane; i < n; i += 64) { const float d = x[i] - mean; v += d*d; }
    v = wave_sum(v);
    const float var = v / n;
    const float sc = 1.0f / sqrtf(var + a.eps);
    for (int i = lane; i < n; i += 64) {
        float r = (x[i] - mean) * sc;
        if (a.w) r = r * a.w[i];
        if (a.b) r = r + a.b[i];
        y[i] = r;
    }
}
// rows of up to 2048 elements (every Whisper width): the row lives in registers (clamped 16-byte loads, all in flight at
// once), one pass over memory instead of three
__global__ void __launch_bounds__(256) k_norm_v4(const NormArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.nrows) return;
    const int64_t i1 = row % a.x.ne[1], i2 = (row / a.x.ne[1]) % a.x.ne[2], i3 = row / (a.x.ne[1]*a.x.ne[2]);
    const float * x = (const float *) (a.x.data + i1*a.x.nb[1] + i2*a.x.nb[2] + i3*a.x.nb[3]);
    float * y = (float *) (a.y.data + i1*a.y.nb[1] + i2*a.y.nb[2] + i3*a.y.nb[3]);
    const int n = (int) a.x.ne[0], n4 = n >> 2;
    float4 xr[8];
    #pragma unroll
    for (int i = 0; i < 8; i++) { const int e4 = lane + 64*i; xr[i] = *(const float4 *) (x + (size_t) (e4 < n4 ? e4 : n4 - 1)*4); }
    // the affine parameters are requested together with the row (clamped, always-valid addresses: x itself stands in when a vector
    // is absent), not after the two reductions: one memory round trip per row instead of two
    const float * wp = a.w ? a.w : x, * bp = a.b ? a.b : x;
    float4 wr[8], br[8];
    #pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e4 = lane + 64*i, c4 = e4 < n4 ? e4 : n4 - 1;
        wr[i] = *(const float4 *) (wp + (size_t) c4*4);
        br[i] = *(const float4 *) (bp + (size_t) c4*4);
    }
    float s = 0.0f;
    #pragma unroll
    for (int i = 0; i < 8; i++) if (lane + 64*i < n4) s += (xr[i].x + xr[i].y) + (xr[i].z + xr[i].w);
    const float mean = wave_sum(s) / n;
    float v = 0.0f;
    #pragma unroll
    for (int i = 0; i < 8; i++) {
        if (lane + 64*i < n4) {
            const float d0 = xr[i].x - mean, d1 = xr[i].y - mean, d2 = xr[i].z - mean, d3 = xr[i].w - mean;
            v += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
        }
    }
    const float sc = 1.0f / sqrtf(wave_sum(v) / n + a.eps);
    #pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e4 = lane + 64*i;
        if (64*i >= n4) break;                               // uniform: no lane of this wave has elements left
        float r[4] = { (xr[i].x - mean) * sc, (xr[i].y - mean) * sc, (xr[i].z - mean) * sc, (xr[i].w - mean) * sc };
        if (a.w) { const float4 w = wr[i]; r[0] = r[0]*w.x; r[1] = r[1]*w.y; r[2] = r[2]*w.z; r[3] = r[3]*w.w; }
        if (a.b) { const float4 b = br[i]; r[0] = r[0]+b.x; r[1] = r[1]+b.y; r[2] = r[2]+b.z; r[3] = r[3]+b.w; }
        if (e4 < n4) *(float4 *) (y + (size_t) e4*4) = make_float4(r[0], r[1], r[2], r[3]);
        if (a.rq) {
            // the activation ROWS of the int8 tile GEMM that consumes this LayerNorm (qrows.h; k_prep_act modes 3 / 4, statement for
            // statement): a Q8_0 block (32 elements) is 8 neighbouring lanes of this iteration, a Q8_K block (256) the whole wave
            uint32_t packed;
            if (a.prep_mode == MI355X_PREP_Q8_0_ROWS) {
                float amax = fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3])));
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                const float d  = amax / 127.0f;
                const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
                const int q0 = (int) rintf(r[0]*id), q1 = (int) rintf(r[1]*id), q2 = (int) rintf(r[2]*id), q3 = (int) rintf(r[3]*id);
                packed = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
                if (e4 < n4 && (e4 & 7) == 0) a.rd[(int64_t) (e4 >> 3) * a.nrows + row] = round_f16(d);
            } else {
                float mx = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
                float mn = fminf(fminf(r[0], r[1]), fminf(r[2], r[3]));
                #pragma unroll
                for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
                const float amax = fmaxf(mx, -mn);
                const float maxv = (mx >= -mn) ? mx : mn;
                int q[4] = { 0, 0, 0, 0 };
                float d = 0.0f;
                if (amax != 0.0f) {
                    const float iscale = -127.0f / maxv;
                    #pragma unroll
                    for (int j = 0; j < 4; j++) { const int qi = (int) rintf(iscale * r[j]); q[j] = qi < 127 ? qi : 127; }
                    d = 1.0f / iscale;
                }
                int sm = q[0] + q[1] + q[2] + q[3];
                sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
                packed = (uint32_t) (q[0] & 0xFF) | ((uint32_t) (q[1] & 0xFF) << 8) | ((uint32_t) (q[2] & 0xFF) << 16) | ((uint32_t) (q[3] & 0xFF) << 24);
                if (e4 < n4 && (e4 & 7) == 0)  a.rs[(int64_t) (e4 >> 3) * a.nrows + row] = sm;
                if (e4 < n4 && (e4 & 63) == 0) a.rd[(int64_t) (e4 >> 6) * a.nrows + row] = d;
            }
            if (e4 < n4) *(uint32_t *) (a.rq + row * n + (size_t) e4*4) = packed;
        } else if (a.prep) {
            // the activation preparation of the MFMA GEMM that consumes this LayerNorm (k_prep_act, gemm_mfma.hip), on the values
            // just stored: same operations in the same order, so the f16 matrix is bit-identical to a separate pass over y.
            // A Q8_0 block (32 elements) is 8 neighbouring lanes of this iteration, a Q8_K block (256) the whole wave.
            float q[4];
            if (a.prep_mode == 0) { q[0] = r[0]; q[1] = r[1]; q[2] = r[2]; q[3] = r[3]; }
            else if (a.prep_mode == 1) {
                float amax = fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3])));
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                const float d  = round_f16(amax / 127.0f);
                const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
                #pragma unroll
                for (int j = 0; j < 4; j++) q[j] = d * rintf(r[j]*id);
            } else {
                float mx = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
                float mn = fminf(fminf(r[0], r[1]), fminf(r[2], r[3]));
                #pragma unroll
                for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
                const float amax = fmaxf(mx, -mn);
                const float maxv = (mx >= -mn) ? mx : mn;
                if (amax == 0.0f) { q[0] = q[1] = q[2] = q[3] = 0.0f; }
                else {
                    const float iscale = -127.0f / maxv;
                    const float d = 1.0f / iscale;
                    #pragma unroll
                    for (int j = 0; j < 4; j++) q[j] = d * fminf(127.0f, rintf(iscale * r[j]));
                }
            }
            if (a.prep_mode != 0) {
                #pragma unroll
                for (int j = 0; j < 4; j++) q[j] = fminf(fmaxf(q[j], -65504.0f), 65504.0f);
            }
            if (e4 < n4) *(uint2 *) (a.prep + row * n + (size_t) e4*4) = make_uint2(f2h(q[0]) | ((uint32_t) f2h(q[1]) << 16), f2h(q[2]) | ((uint32_t) f2h(q[3]) << 16));
        }
    }
}
// LayerNorm that also leaves the prepared f16 [T][K] activation matrix of the GEMM consuming it (mi355x_prep_act's result on dst)
extern "C" int mi355x_norm_prep(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * y, float eps, const float * w, const float * b,
                                void * prep, int mode) {
    const int64_t K = x->ne[0];
    if (x->type != MI355X_TYPE_F32 || y->type != MI355X_TYPE_F32 || x->nb[0] != 4 || y->nb[0] != 4 || !t_same_shape(x, y)) return MI355X_E_UNSUPPORTED;
    if (!prep || mode < 0 || mode > 4 || x->ne[2] != 1 || x->ne[3] != 1 || K % 4 || K > 2048 || (mode == 1 && K % 32) || (mode == 2 && K % 256)) return MI355X_E_UNSUPPORTED;
    if (mode >= 3 && (K % 128 || (mode == 4 && K % 256))) return MI355X_E_UNSUPPORTED;
    if (((uintptr_t) x->data | (uintptr_t) y->data | (uintptr_t) w | (uintptr_t) b | (uintptr_t) prep) % 16 || (x->nb[1] | y->nb[1]) % 16) return MI355X_E_UNSUPPORTED;
    NormArgs k = { to_d(x), to_d(y), eps, w, b, t_nrows(x), (uint16_t *) prep, mode, nullptr, nullptr, nullptr };
    if (mode >= 3) { const qrows_t R = qrows_of(prep, mode == 4, K, k.nrows); k.rq = R.q; k.rd = R.d; k.rs = R.bsum; k.prep = nullptr; }
    if (k.nrows == 0 || K == 0) return 0;
    return emit(ctx, "norm_prep", k_norm_v4, dim3((uint32_t) ((k.nrows + 3) / 4)), dim3(256), 0, k, (double) t_nelements(x) * 10, 0);
}

extern "C" int mi355x_norm(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * y, float eps, const float * w, const float * b) {
    if (x->type != MI355X_TYPE_F32 || y->type != MI355X_TYPE_F32 || x->nb[0] != 4 || y->nb[0] != 4 || !t_same_shape(x, y)) return MI355X_E_UNSUPPORTED;
    NormArgs k = { to_d(x), to_d(y), eps, w, b, t_nrows(x), nullptr, 0 };
    if (k.nrows == 0 || x->ne[0] == 0) return 0;
    const bool v4 = x->ne[0] % 4 == 0 && x->ne[0] <= 2048 && ((uintptr_t) x->data | (uintptr_t) y->data | (uintptr_t) w | (uintptr_t) b) % 16 == 0 &&
                    (x->nb[1] | x->nb[2] | x->nb[3] | y->nb[1] | y->nb[2] | y->nb[3]) % 16 == 0;
    if (v4) return emit(ctx, "norm", k_norm_v4, dim3((uint32_t) ((k.nrows + 3) / 4)), dim3(256), 0, k, (double) t_nelements(x) * 8, 0);
    return emit(ctx, "norm", k_norm, dim3((uint32_t) ((k.nrows + 3) / 4)), dim3(256), 0, k, (double) t_nelements(x) * 8, 0);
}

// -------------------------------------------------------------------------------------------------
// cpy / cont / dup / cast between F32 and F16 (ggml-cpu/ops.cpp:17-654): element i of src (in src's
// logical order) goes to element i of dst (in dst's logical order); shapes may differ, counts equal.
// -------------------------------------------------------------------------------------------------
struct CpyArgs { dtensor s, d; int64_t n; int st, dt; };
template <int ST, int DT>
__global__ void __launch_bounds__(256) k_cpy_generic(const CpyArgs a) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t) gridDim.x * 256) {
        int64_t r = i;
        const int64_t s0 = r % a.s.ne[0]; r /= a.s.ne[0];
        const int64_t s1 = r % a.s.ne[1]; r /= a.s.ne[1];
        const int64_t s2 = r % a.s.ne[2]; const int64_t s3 = r / a.s.ne[2];
        r = i;
        const int64_t d0 = r % a.d.ne[0]; r /= a.d.ne[0];
        const int64_t d1 = r % a.d.ne[1]; r /= a.d.ne[1];
        const int64_t d2 = r % a.d.ne[2]; const int64_t d3 = r / a.d.ne[2];
        const char * ps = a.s.data + s0*a.s.nb[0] + s1*a.s.nb[1] + s2*a.s.nb[2] + s3*a.s.nb[3];
        char * pd = a.d.data + d0*a.d.nb[0] + d1*a.d.nb[1] + d2*a.d.nb[2] + d3*a.d.nb[3];
        float v;
        if (ST == MI355X_TYPE_F32) v = *(const float *) ps; else v = h2f(*(const uint16_t *) ps);
        if (DT == MI355X_TYPE_F32) *(float *) pd = v;
        else if (ST == MI355X_TYPE_F16) *(uint16_t *) pd = *(const uint16_t *) ps;
        else *(uint16_t *) pd = f2h(v);
    }
}
// contiguous fast path, 4 elements per thread
struct CpyFastArgs { const void * s; void * d; int64_t n; };
template <int ST, int DT>
__global__ void __launch_bounds__(256) k_cpy_contig(const CpyFastArgs a) {
    for (int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (int64_t) gridDim.x * 1024) {
        float v[4]; uint16_t hraw[4];
        const int m = (int) (a.n - i < 4 ? a.n - i : 4);
        if (ST == MI355X_TYPE_F32) {
            const float * s = (const float *) a.s + i;
            if (m == 4) { const float4 x = *(const float4 *) s; v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
            else for (int j = 0; j < m; j++) v[j] = s[j];
        } else {
            const uint16_t * s = (const uint16_t *) a.s + i;
            if (m == 4) { const uint2 x = *(const uint2 *) s; hraw[0] = x.x & 0xFFFF; hraw[1] = x.x >> 16; hraw[2] = x.y & 0xFFFF; hraw[3] = x.y >> 16; }
            else for (int j = 0; j < m; j++) hraw[j] = s[j];
            for (int j = 0; j < m; j++) v[j] = h2f(hraw[j]);
        }
        if (DT == MI355X_TYPE_F32) {
            float * d = (float *) a.d + i;
            if (m == 4) *(float4 *) d = make_float4(v[0], v[1], v[2], v[3]);
            else for (int j = 0; j < m; j++) d[j] = v[j];
        } else {
            uint16_t * d = (uint16_t *) a.d + i;
            uint16_t h[4];
            for (int j = 0; j < m; j++) h[j] = ST == MI355X_TYPE_F16 ? hraw[j] : f2h(v[j]);
            if (m == 4) *(uint2 *) d = make_uint2(h[0] | ((uint32_t) h[1] << 16), h[2] | ((uint32_t) h[3] << 16));
            else for (int j = 0; j < m; j++) d[j] = h[j];
        }
    }
}
// F32 matrix transpose through a 64 x 64 LDS tile: the ggml_cont(ggml_transpose(x)) of src/whisper.cpp:2069 ([1500 x 1280] floats per
// encode).  The generic kernel reads such a source one dword per 128-byte line (PMC: 62 MB fetched for 7.7 MB written).
struct TransposeArgs { const float * s; float * d; int64_t ld; int n0, n1; };     // d[i1][i0] = s[i0*ld + i1], i0 < n0, i1 < n1
__global__ void __launch_bounds__(256) k_transpose_f32(const TransposeArgs a) {
    __shared__ float tile[64][65];
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    const int b1 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    #pragma unroll
    for (int r = r0; r < 64; r += 4)
        if (b0 + r < a.n0 && b1 + c < a.n1) tile[r][c] = a.s[(int64_t) (b0 + r) * a.ld + b1 + c];
    __syncthreads();
    #pragma unroll
    for (int r = r0; r < 64; r += 4)
        if (b1 + r < a.n1 && b0 + c < a.n0) a.d[(int64_t) (b1 + r) * a.n0 + b0 + c] = tile[c][r];
}

extern "C" int mi355x_cpy(mi355x_ctx * ctx, const mi355x_tensor * s, const mi355x_tensor * d) {
    const int st = s->type, dt = d->type;
    if ((st != MI355X_TYPE_F32 && st != MI355X_TYPE_F16) || (dt != MI355X_TYPE_F32 && dt != MI355X_TYPE_F16)) return MI355X_E_UNSUPPORTED;
    const int64_t n = t_nelements(s);
    if (n != t_nelements(d)) return MI355X_E_UNSUPPORTED;
    if (n == 0) return 0;
    const double bytes = (double) n * ((st == MI355X_TYPE_F32 ? 4 : 2) + (dt == MI355X_TYPE_F32 ? 4 : 2));
    const int64_t nb4 = (n + 1023) / 1024;
    if (t_is_contiguous(s) && t_is_contiguous(d) && ((uintptr_t) s->data % 16 == 0) && ((uintptr_t) d->data % 16 == 0)) {
        CpyFastArgs k = { s->data, d->data, n };
        const dim3 g((uint32_t) (nb4 < 8192 ? nb4 : 8192));
        if (st == MI355X_TYPE_F32 && dt == MI355X_TYPE_F32) return emit(ctx, "cpy", k_cpy_contig<MI355X_TYPE_F32, MI355X_TYPE_F32>, g, dim3(256), 0, k, bytes, 0);
        if (st == MI355X_TYPE_F32 && dt == MI355X_TYPE_F16) return emit(ctx, "cpy", k_cpy_contig<MI355X_TYPE_F32, MI355X_TYPE_F16>, g, dim3(256), 0, k, bytes, 0);
        if (st == MI355X_TYPE_F16 && dt == MI355X_TYPE_F32) return emit(ctx, "cpy", k_cpy_contig<MI355X_TYPE_F16, MI355X_TYPE_F32>, g, dim3(256), 0, k, bytes, 0);
        return emit(ctx, "cpy", k_cpy_contig<MI355X_TYPE_F16, MI355X_TYPE_F16>, g, dim3(256), 0, k, bytes, 0);
    }
    // a transposed 2-D F32 source (nb[1] == 4: rows of the underlying matrix run along dim 1) into a contiguous destination of the same shape
    if (st == MI355X_TYPE_F32 && dt == MI355X_TYPE_F32 && s->ne[2] == 1 && s->ne[3] == 1 && s->nb[1] == 4 && s->nb[0] % 4 == 0 && s->nb[0] >= s->ne[1]*4 &&
        s->ne[0] >= 16 && s->ne[1] >= 16 && d->ne[0] == s->ne[0] && d->ne[1] == s->ne[1] && t_is_contiguous(d) && (s->ne[0] + 63) / 64 <= 65535) {
        TransposeArgs k = { (const float *) s->data, (float *) d->data, s->nb[0] / 4, (int) s->ne[0], (int) s->ne[1] };
        return emit(ctx, "transpose", k_transpose_f32, dim3((uint32_t) ((s->ne[1] + 63) / 64), (uint32_t) ((s->ne[0] + 63) / 64)), dim3(256), 0, k, bytes, 0);
    }
    CpyArgs k = { to_d(s), to_d(d), n, st, dt };
    const int64_t nb = (n + 255) / 256;
    const dim3 g((uint32_t) (nb < 16384 ? nb : 16384));
    if (st == MI355X_TYPE_F32 && dt == MI355X_TYPE_F32) return emit(ctx, "cpy", k_cpy_generic<MI355X_TYPE_F32, MI355X_TYPE_F32>, g, dim3(256), 0, k, bytes, 0);
    if (st == MI355X_TYPE_F32 && dt == MI355X_TYPE_F16) return emit(ctx, "cpy", k_cpy_generic<MI355X_TYPE_F32, MI355X_TYPE_F16>, g, dim3(256), 0, k, bytes, 0);
    if (st == MI355X_TYPE_F16 && dt == MI355X_TYPE_F32) return emit(ctx, "cpy", k_cpy_generic<MI355X_TYPE_F16, MI355X_TYPE_F32>, g, dim3(256), 0, k, bytes, 0);
    return emit(ctx, "cpy", k_cpy_generic<MI355X_TYPE_F16, MI355X_TYPE_F16>, g, dim3(256), 0, k, bytes, 0);
}

// -------------------------------------------------------------------------------------------------
// get_rows (ggml-cpu/ops.cpp:4850-5017): dst[:, i10, i11, i12] = src[:, idx[i10,i11,i12], i11, i12]
// -------------------------------------------------------------------------------------------------
// optional fused second gather + add (token embedding + positional embedding, src/whisper.cpp:2524-2526):
//   dst[:, r] = src[:, idx[r]] + add[:, add_idx[r]]     (add: F32 rows, 1-D index of the same length)
struct GetRowsArgs { dtensor s, idx, d; int type; int64_t nbt; const char * add; int64_t add_nb1; const int32_t * add_idx; int64_t add_rows; };
template <int TYPE>
__global__ void __launch_bounds__(256) k_get_rows(const GetRowsArgs a) {
    const int64_t r = blockIdx.x;     // flattened (i10, i11, i12)
    const int64_t i10 = r % a.idx.ne[0], i11 = (r / a.idx.ne[0]) % a.idx.ne[1], i12 = r / (a.idx.ne[0]*a.idx.ne[1]);
    const int32_t row = *(const int32_t *) (a.idx.data + i10*a.idx.nb[0] + i11*a.idx.nb[1] + i12*a.idx.nb[2]);
    float * dst = (float *) (a.d.data + i10*a.d.nb[1] + i11*a.d.nb[2] + i12*a.d.nb[3]);
    const int64_t ne0 = a.s.ne[0];
    if (row < 0 || row >= a.s.ne[1]) return;
    const float * addrow = nullptr;
    if (a.add) {
        const int32_t ar = a.add_idx[r];
        if (ar < 0 || ar >= a.add_rows) return;
        addrow = (const float *) (a.add + (int64_t) ar*a.add_nb1);
    }
    if constexpr (TYPE == MI355X_TYPE_F32 || TYPE == MI355X_TYPE_F16) {
        const char * src = a.s.data + (int64_t) row*a.s.nb[1] + i11*a.s.nb[2] + i12*a.s.nb[3];
        for (int64_t i = threadIdx.x; i < ne0; i += 256) {
            const float v = TYPE == MI355X_TYPE_F32 ? ((const float *) src)[i] : h2f(((const uint16_t *) src)[i]);
            dst[i] = addrow ? v + addrow[i] : v;
        }
    } else {
        const qplanes<TYPE> p(a.s.data, a.nbt);
        const int64_t nb32 = ne0 / 32;                     // 32-element groups in a row
        for (int64_t g = threadIdx.x; g < nb32; g += 256) {
            float v[32];
            dequant_block32<TYPE>(p, (int64_t) row * nb32 + g, v);
            if (addrow) {
                #pragma unroll
                for (int j = 0; j < 32; j += 4) { const float4 b = *(const float4 *) (addrow + g*32 + j); v[j] += b.x; v[j+1] += b.y; v[j+2] += b.z; v[j+3] += b.w; }
            }
            #pragma unroll
            for (int j = 0; j < 32; j += 4) *(float4 *) (dst + g*32 + j) = make_float4(v[j], v[j+1], v[j+2], v[j+3]);
        }
    }
}
static int get_rows_impl(mi355x_ctx * ctx, const mi355x_tensor * s, const mi355x_tensor * idx, const mi355x_tensor * d,
                         const mi355x_tensor * add, const mi355x_tensor * add_idx);
extern "C" int mi355x_get_rows(mi355x_ctx * ctx, const mi355x_tensor * s, const mi355x_tensor * idx, const mi355x_tensor * d) {
    return get_rows_impl(ctx, s, idx, d, nullptr, nullptr);
}
extern "C" int mi355x_get_rows_add(mi355x_ctx * ctx, const mi355x_tensor * s, const mi355x_tensor * idx,
                                   const mi355x_tensor * add, const mi355x_tensor * add_idx, const mi355x_tensor * d) {
    if (!add || !add_idx || add->type != MI355X_TYPE_F32 || add->nb[0] != 4 || add->ne[0] != s->ne[0] || add->ne[2] != 1 || add->ne[3] != 1 ||
        add_idx->type != MI355X_TYPE_I32 || add_idx->nb[0] != 4 || add_idx->ne[1] != 1 || add_idx->ne[2] != 1 || idx->ne[1] != 1 || idx->ne[2] != 1 ||
        add_idx->ne[0] != idx->ne[0] || ((uintptr_t) add->data % 16) || (add->nb[1] % 16)) return MI355X_E_UNSUPPORTED;
    return get_rows_impl(ctx, s, idx, d, add, add_idx);
}
static int get_rows_impl(mi355x_ctx * ctx, const mi355x_tensor * s, const mi355x_tensor * idx, const mi355x_tensor * d,
                         const mi355x_tensor * add, const mi355x_tensor * add_idx) {
    if (idx->type != MI355X_TYPE_I32 || d->type != MI355X_TYPE_F32 || d->nb[0] != 4) return MI355X_E_UNSUPPORTED;
    const int64_t nr = idx->ne[0]*idx->ne[1]*idx->ne[2];
    if (nr == 0) return 0;
    GetRowsArgs k = { to_d(s), to_d(idx), to_d(d), s->type, 0, nullptr, 0, nullptr, 0 };
    if (add) { k.add = (const char *) add->data; k.add_nb1 = add->nb[1]; k.add_idx = (const int32_t *) add_idx->data; k.add_rows = add->ne[1]; }
    const double bytes = (double) nr * (mi355x_type_row_bytes(s->type, s->ne[0]) + s->ne[0]*4.0);
    const dim3 g((uint32_t) nr), b(256);
    if (mi355x_type_is_quantized(s->type)) {
        if (!t_is_contiguous(s) || s->ne[2] != 1 || s->ne[3] != 1 || ((uintptr_t) d->data % 16) || (d->nb[1] % 16)) return MI355X_E_UNSUPPORTED;
        k.nbt = t_nelements(s) / type_block(s->type);
    }
    switch (s->type) {
        case MI355X_TYPE_F32:  if (s->nb[0] != 4) return MI355X_E_UNSUPPORTED; return emit(ctx, "get_rows", k_get_rows<MI355X_TYPE_F32>,  g, b, 0, k, bytes, 0);
        case MI355X_TYPE_F16:  if (s->nb[0] != 2) return MI355X_E_UNSUPPORTED; return emit(ctx, "get_rows", k_get_rows<MI355X_TYPE_F16>,  g, b, 0, k, bytes, 0);
        case MI355X_TYPE_Q4_0: return emit(ctx, "get_rows", k_get_rows<MI355X_TYPE_Q4_0>, g, b, 0, k, bytes, 0);
        case MI355X_TYPE_Q5_0: return emit(ctx, "get_rows", k_get_rows<MI355X_TYPE_Q5_0>, g, b, 0, k, bytes, 0);
        case MI355X_TYPE_Q8_0: return emit(ctx, "get_rows", k_get_rows<MI355X_TYPE_Q8_0>, g, b, 0, k, bytes, 0);
        case MI355X_TYPE_Q4_K: return emit(ctx, "get_rows", k_get_rows<MI355X_TYPE_Q4_K>, g, b, 0, k, bytes, 0);
        default: return MI355X_E_UNSUPPORTED;
    }
}

// -------------------------------------------------------------------------------------------------
// im2col, 1-D (ggml-cpu/ops.cpp:6437-6517): dst[(n*OW + ow)*(IC*KW) + ic*KW + k] = f16(x[n][ic][ow*s0 + k*d0 - p0]) or 0
// thread per (ow, ic): writes KW contiguous outputs; lanes along ow so the x reads are coalesced.
// -------------------------------------------------------------------------------------------------
struct Im2colArgs { dtensor x; char * dst; int dst_f16; int64_t IW, IC, N, OW; int KW, s0, p0, d0; };
__global__ void __launch_bounds__(256) k_im2col_1d(const Im2colArgs a) {
    const int64_t ow = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t ic = blockIdx.y, n = blockIdx.z;
    if (ow >= a.OW) return;
    const float * x = (const float *) (a.x.data + ic*a.x.nb[1] + n*a.x.nb[2]);
    const int64_t o = ((n*a.OW + ow) * a.IC + ic) * a.KW;
    for (int k = 0; k < a.KW; k++) {
        const int64_t iw = ow*a.s0 + (int64_t) k*a.d0 - a.p0;
        const float v = (iw < 0 || iw >= a.IW) ? 0.0f : x[iw];
        if (a.dst_f16) ((uint16_t *) a.dst)[o + k] = f2h(v); else ((float *) a.dst)[o + k] = v;
    }
}
extern "C" int mi355x_im2col_1d(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * d, int kw, int s0, int p0, int d0) {
    if (x->type != MI355X_TYPE_F32 || x->nb[0] != 4 || (d->type != MI355X_TYPE_F16 && d->type != MI355X_TYPE_F32) || !t_is_contiguous(d)) return MI355X_E_UNSUPPORTED;
    Im2colArgs k = { to_d(x), (char *) d->data, d->type == MI355X_TYPE_F16, x->ne[0], x->ne[1], x->ne[2], d->ne[1], kw, s0, p0, d0 };
    if (d->ne[0] != k.IC*kw || d->ne[2] != k.N || k.IC > 65535 || k.N > 65535) return MI355X_E_UNSUPPORTED;
    if (t_nelements(d) == 0) return 0;
    return emit(ctx, "im2col", k_im2col_1d, dim3((uint32_t) ((k.OW + 255) / 256), (uint32_t) k.IC, (uint32_t) k.N), dim3(256), 0, k,
                (double) t_nelements(x)*4 + (double) t_nelements(d) * (k.dst_f16 ? 2 : 4), 0);
}

// -------------------------------------------------------------------------------------------------
// soft_max_ext (ggml-cpu/ops.cpp:5455-5565): y = softmax(x*scale + slope*mask), one wave per row
// -------------------------------------------------------------------------------------------------
struct SoftmaxArgs { dtensor x, m, y; int has_mask, mask_f16; float scale, max_bias, m0, m1; uint32_t n_head_log2; int64_t nrows; };
__global__ void __launch_bounds__(256) k_soft_max(const SoftmaxArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.nrows) return;
    const int64_t i1 = row % a.x.ne[1], i2 = (row / a.x.ne[1]) % a.x.ne[2], i3 = row / (a.x.ne[1]*a.x.ne[2]);
    const float * x = (const float *) (a.x.data + i1*a.x.nb[1] + i2*a.x.nb[2] + i3*a.x.nb[3]);
    float * y = (float *) (a.y.data + i1*a.y.nb[1] + i2*a.y.nb[2] + i3*a.y.nb[3]);
    const char * mp = a.has_mask ? a.m.data + i1*a.m.nb[1] + (i2 % a.m.ne[2])*a.m.nb[2] + (i3 % a.m.ne[3])*a.m.nb[3] : nullptr;
    float slope = 1.0f;
    if (a.max_bias > 0.0f) {
        const uint32_t h = (uint32_t) i2;
        slope = h < a.n_head_log2 ? powf(a.m0, (float) (h + 1)) : powf(a.m1, (float) (2*(h - a.n_head_log2) + 1));
    }
    const int n = (int) a.x.ne[0];
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        float w = x[i] * a.scale;
        if (mp) w += slope * (a.mask_f16 ? h2f(((const uint16_t *) mp)[i]) : ((const float *) mp)[i]);
        mx = fmaxf(mx, w);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int i = lane; i < n; i += 64) {
        float w = x[i] * a.scale;
        if (mp) w += slope * (a.mask_f16 ? h2f(((const uint16_t *) mp)[i]) : ((const float *) mp)[i]);
        const float e = expf(w - mx);
        y[i] = e; sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < n; i += 64) y[i] *= inv;
}
extern "C" int mi355x_soft_max(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * mask, const mi355x_tensor * y, float scale, float max_bias) {
    if (x->type != MI355X_TYPE_F32 || y->type != MI355X_TYPE_F32 || x->nb[0] != 4 || y->nb[0] != 4 || !t_same_shape(x, y)) return MI355X_E_UNSUPPORTED;
    if (mask && ((mask->type != MI355X_TYPE_F32 && mask->type != MI355X_TYPE_F16) || mask->ne[0] != x->ne[0] || mask->ne[1] < x->ne[1])) return MI355X_E_UNSUPPORTED;
    SoftmaxArgs k; memset(&k, 0, sizeof(k));
    k.x = to_d(x); k.y = to_d(y); if (mask) k.m = to_d(mask);
    k.has_mask = mask != nullptr; k.mask_f16 = mask && mask->type == MI355X_TYPE_F16;
    k.scale = scale; k.max_bias = max_bias;
    const uint32_t n_head = (uint32_t) x->ne[2];
    k.n_head_log2 = 1u << (uint32_t) floor(log2((double) n_head));
    k.m0 = powf(2.0f, -(max_bias) / k.n_head_log2); k.m1 = powf(2.0f, -(max_bias / 2.0f) / k.n_head_log2);
    k.nrows = t_nrows(x);
    if (k.nrows == 0 || x->ne[0] == 0) return 0;
    return emit(ctx, "soft_max", k_soft_max, dim3((uint32_t) ((k.nrows + 3) / 4)), dim3(256), 0, k, (double) t_nelements(x) * 8, 0);
}

// -------------------------------------------------------------------------------------------------
// rope, modes NORMAL (0) and NEOX (2), f32 (ggml-cpu/ops.cpp:5822-6131).  One thread per rotated pair.
// theta for pair p: pos * theta_scale^p accumulated by repeated multiplication exactly as
// ggml_rope_cache_init does (sequential product, so the rounding matches the CPU cache).
// -------------------------------------------------------------------------------------------------
struct RopeArgs { dtensor x, y; const int32_t * pos; const float * ff; mi355x_rope_params p; float theta_scale, corr0, corr1; };
__global__ void __launch_bounds__(64) k_rope(const RopeArgs a) {
    // block = one row (i1 = head, i2 = position, i3 = batch); lane loops over pairs
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.x.ne[1], i2 = (row / a.x.ne[1]) % a.x.ne[2], i3 = row / (a.x.ne[1]*a.x.ne[2]);
    const float * x = (const float *) (a.x.data + i1*a.x.nb[1] + i2*a.x.nb[2] + i3*a.x.nb[3]);
    float * y = (float *) (a.y.data + i1*a.y.nb[1] + i2*a.y.nb[2] + i3*a.y.nb[3]);
    const int ne0 = (int) a.x.ne[0], n_dims = a.p.n_dims;
    const int lane = threadIdx.x;
    // each lane computes its theta by the same sequential product the CPU uses: theta_p = (((pos*ts)*ts)...)
    // modes (ggml.h:250-254): 0 NORMAL (adjacent pairs), 2 NEOX (pairs n_dims/2 apart), 8 MROPE / 40 IMROPE (four position streams t, h, w, e
    // chosen per pair by `sections`, NEOX pairing), 24 VISION (sections restart their own theta, pairs n_dims apart over the whole row) —
    // ggml_mrope_cache_init + rotate_pairs, ggml-cpu/ops.cpp:5868-5945
    const int mode = a.p.mode;
    const bool mrope = (mode & 8) != 0, vision = mode == 24, imrope = mode == 40;
    const int s0 = a.p.sections[0], s1 = a.p.sections[1], s2 = a.p.sections[2], s3 = a.p.sections[3];
    const int sect_dims = s0 + s1 + s2 + s3, sec_w = s1 + s0, sec_e = s2 + sec_w;
    const int64_t ne2 = a.x.ne[2];
    for (int pr = lane; pr < ne0/2; pr += 64) {
        const int i0 = 2*pr;
        if (vision || i0 < n_dims) {
            float theta; int nmul = pr;
            if (!mrope) theta = (float) a.pos[i2];
            else {
                const int sector = pr % sect_dims;
                int which = 0;                   // 0 t, 1 h, 2 w, 3 e
                if (imrope) {
                    if (sector % 3 == 1 && sector < 3*s1) which = 1; else if (sector % 3 == 2 && sector < 3*s2) which = 2; else if (sector % 3 == 0 && sector < 3*s0) which = 0; else which = 3;
                } else {
                    if (sector >= s0 && sector < sec_w) which = 1; else if (sector >= sec_w && sector < sec_w + s2) which = 2; else if (sector >= sec_w + s2) which = 3;
                }
                theta = (float) a.pos[i2 + (int64_t) which*ne2];
                // VISION restarts each section's theta where the section begins (indep_sects): multiplications since that restart
                if (vision) nmul = sector - (which == 0 ? 0 : (which == 1 ? s0 : (which == 2 ? sec_w : sec_e)));
            }
            for (int q = 0; q < nmul; q++) theta *= a.theta_scale;
            const float ffv = a.ff ? a.ff[pr] : 1.0f;
            const float theta_extrap = theta / ffv;
            float theta_interp = a.p.freq_scale * theta_extrap;
            float th = theta_interp, mscale = a.p.attn_factor;
            if (a.p.ext_factor != 0.0f) {
                const float yv = ((float) (i0 / 2) - a.corr0) / fmaxf(0.001f, a.corr1 - a.corr0);
                const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, yv))) * a.p.ext_factor;
                th = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
                mscale *= 1.0f + 0.1f * logf(1.0f / a.p.freq_scale);
            }
            const float c = cosf(th) * mscale, s = sinf(th) * mscale;
            int ia, ib;
            if (mode == 0) { ia = i0; ib = i0 + 1; } else if (vision) { ia = pr; ib = pr + n_dims; } else { ia = pr; ib = pr + n_dims/2; }
            const float x0 = x[ia], x1 = x[ib];
            y[ia] = x0*c - x1*s;
            y[ib] = x0*s + x1*c;
        } else {
            y[i0] = x[i0]; y[i0 + 1] = x[i0 + 1];
        }
    }
}
extern "C" int mi355x_rope(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * pos, const float * ff, const mi355x_tensor * y, const mi355x_rope_params * p) {
    if (x->type != MI355X_TYPE_F32 || y->type != MI355X_TYPE_F32 || x->nb[0] != 4 || y->nb[0] != 4 || !t_same_shape(x, y)) return MI355X_E_UNSUPPORTED;
    const bool mrope = (p->mode & 8) != 0;
    if (p->mode != 0 && p->mode != 2 && p->mode != 8 && p->mode != 24 && p->mode != 40) return MI355X_E_UNSUPPORTED;
    if (pos->type != MI355X_TYPE_I32 || pos->ne[0] < x->ne[2] * (mrope ? 4 : 1) || (p->n_dims & 1) || p->n_dims > x->ne[0] || (x->ne[0] & 1)) return MI355X_E_UNSUPPORTED;
    if (mrope && (p->sections[0] < 0 || p->sections[1] < 0 || p->sections[2] < 0 || p->sections[3] < 0 || p->sections[0] + p->sections[1] + p->sections[2] + p->sections[3] <= 0 ||
                  p->sections[0] + p->sections[1] + p->sections[2] + p->sections[3] > x->ne[0])) return MI355X_E_UNSUPPORTED;
    if (p->mode == 24 && p->n_dims != x->ne[0] / 2) return MI355X_E_UNSUPPORTED;
    RopeArgs k; k.x = to_d(x); k.y = to_d(y); k.pos = (const int32_t *) pos->data; k.ff = ff; k.p = *p;
    k.theta_scale = powf(p->freq_base, -2.0f / p->n_dims);
    // ggml_rope_yarn_corr_dims (ggml/src/ggml.c:4371-4383)
    auto corr_dim = [&](float n_rot) { return p->n_dims * logf(p->n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(p->freq_base)); };
    const float start = floorf(corr_dim(p->beta_fast)), end = ceilf(corr_dim(p->beta_slow));
    k.corr0 = fmaxf(0.0f, start); k.corr1 = fminf((float) (p->n_dims - 1), end);
    const int64_t nr = t_nrows(x);
    if (nr == 0) return 0;
    return emit(ctx, "rope", k_rope, dim3((uint32_t) nr), dim3(64), 0, k, (double) t_nelements(x) * 8, 0);
}

// -------------------------------------------------------------------------------------------------
// arg-max of a logits row with the runner-up's value (device-side greedy sampling: src/whisper.cpp:6486-6543 picks the most probable
// token of the row the decoder left in HBM).  One workgroup; the FIRST maximum wins, like the sampler's strict `>` scan.
// -------------------------------------------------------------------------------------------------
struct ArgmaxArgs { const float * x; int n; int * out; };
__global__ void __launch_bounds__(1024) k_argmax_top2(const ArgmaxArgs a) {
    __shared__ float s1[16], s2[16]; __shared__ int si[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float b1 = -INFINITY, b2 = -INFINITY; int bi = 0x7FFFFFFF;
    for (int i = tid; i < a.n; i += 1024) {
        const float v = a.x[i];
        if (v > b1) { b2 = b1; b1 = v; bi = i; } else if (v > b2) b2 = v;
    }
    // merge (b1, bi, b2) across lanes: larger value wins, equal values -> smaller index
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float o1 = __shfl_xor(b1, o, 64), o2 = __shfl_xor(b2, o, 64); const int oi = __shfl_xor(bi, o, 64);
        const bool take = o1 > b1 || (o1 == b1 && oi < bi);
        const float lose = take ? b1 : o1;
        b2 = fmaxf(fmaxf(b2, o2), lose);
        if (take) { b1 = o1; bi = oi; }
    }
    if (lane == 0) { s1[wave] = b1; s2[wave] = b2; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        float r1 = s1[0], r2 = s2[0]; int ri = si[0];
        for (int w = 1; w < 16; w++) {
            const bool take = s1[w] > r1 || (s1[w] == r1 && si[w] < ri);
            const float lose = take ? r1 : s1[w];
            r2 = fmaxf(fmaxf(r2, s2[w]), lose);
            if (take) { r1 = s1[w]; ri = si[w]; }
        }
        a.out[0] = ri; ((float *) a.out)[1] = r1; ((float *) a.out)[2] = r2; a.out[3] = a.n;
    }
}
extern "C" int mi355x_argmax_top2(mi355x_ctx * ctx, const float * x_dev, int n, void * out16) {
    if (!x_dev || n < 1 || !out16 || ((uintptr_t) x_dev % 4)) return MI355X_E_UNSUPPORTED;
    ArgmaxArgs k = { x_dev, n, (int *) out16 };
    return emit(ctx, "argmax_top2", k_argmax_top2, dim3(1), dim3(1024), 0, k, (double) n * 4, 0);
}

// -------------------------------------------------------------------------------------------------
// concat f32 along dim
// -------------------------------------------------------------------------------------------------
struct ConcatArgs { dtensor a, b, d; int dim; int64_t n; };
__global__ void __launch_bounds__(256) k_concat(const ConcatArgs a) {
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t) gridDim.x * 256) {
        int64_t r = i, idx[4];
        idx[0] = r % a.d.ne[0]; r /= a.d.ne[0]; idx[1] = r % a.d.ne[1]; r /= a.d.ne[1]; idx[2] = r % a.d.ne[2]; idx[3] = r / a.d.ne[2];
        float * o = (float *) (a.d.data + idx[0]*a.d.nb[0] + idx[1]*a.d.nb[1] + idx[2]*a.d.nb[2] + idx[3]*a.d.nb[3]);
        const dtensor * s = &a.a;
        if (idx[a.dim] >= a.a.ne[a.dim]) { idx[a.dim] -= a.a.ne[a.dim]; s = &a.b; }
        *o = *(const float *) (s->data + idx[0]*s->nb[0] + idx[1]*s->nb[1] + idx[2]*s->nb[2] + idx[3]*s->nb[3]);
    }
}
extern "C" int mi355x_concat(mi355x_ctx * ctx, const mi355x_tensor * a, const mi355x_tensor * b, const mi355x_tensor * d, int dim) {
    if (a->type != MI355X_TYPE_F32 || b->type != MI355X_TYPE_F32 || d->type != MI355X_TYPE_F32 || dim < 0 || dim > 3) return MI355X_E_UNSUPPORTED;
    ConcatArgs k = { to_d(a), to_d(b), to_d(d), dim, t_nelements(d) };
    if (k.n == 0) return 0;
    const int64_t nb = (k.n + 255) / 256;
    return emit(ctx, "concat", k_concat, dim3((uint32_t) (nb < 16384 ? nb : 16384)), dim3(256), 0, k, (double) k.n * 8, 0);
}
