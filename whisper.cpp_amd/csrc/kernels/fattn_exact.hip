// Reference-exact flash attention (opt-in: GGML_MI355X_EXACT=1 in the backend, mi355x_flash_attn_ext_exact in the C ABI).
//
// The production attention kernels (fattn.hip, decode.hip) accumulate in F32 and are CLOSER to the exact softmax(QK)V than
// the reference CPU path is.  The CPU path, however, is what "parity" is measured against, and its arithmetic depends on the
// shape (ggml-cpu/ops.cpp:9077-9230):
//   T == 1 and n_kv >= 512 : split-KV over the thread count: every thread runs the vec path over ceil(n_kv / nth) keys, F32
//                            partials merged sequentially (ops.cpp:8992-9075).  Decoder cross-attention of every single-token step.
//   T >= 64                : tiled F32 path (ops.cpp:8717-8990): q NOT rounded to f16, 64-key tiles, ggml_v_expf, F32 accumulation.
//                            Encoder self-attention, prompts.
//   otherwise              : vec path (ops.cpp:8479-8715): q -> f16, ONE sequential pass over the keys with the running output
//                            kept in F16 (a 2^-11 rounding per key and dim: ~3e-5 NMSE after 1536 keys).  Decoder self-attention,
//                            cross-attention of 2..63-token steps (beam search).
// These kernels walk the same path in the same order with the same roundings (one wave per (head, query[, chunk]); the key loop
// is sequential like the CPU's), so that a free-running greedy / beam decode can be compared token for token with the CPU
// reference.  Scores follow the AVX2 lane order of ggml_vec_dot_f16 (vec.cpp:264; reduction simd-mappings.h:602-620), exp is a
// restatement of glibc's expf (double arithmetic, 2^(i/32) table: bit-identical to libm on 40M random arguments but two).
// The restatement these kernels are checked against is oracle/oracle.c:oracle_flash_attn_ext, itself bit-identical to the
// reference build on the golden cases (tests/test_oracle.py).  Speed is not a goal here.
#include "common.h"
#include <math.h>

struct FXArgs {
    dtensor q, k, v, m, d;
    int has_mask; float scale;
    int T, n_kv, H, rk2, rv2;
    int chunk, nchunks;            // vec path: keys per chunk, number of chunks (1 unless split-KV)
    float * part;                  // split-KV partial records [H][nchunks][66] = {M, S, vkq[64]}
};

// ---- expf as glibc computes it (sysdeps/ieee754/flt-32/e_expf.c, N = 32): everything in double, one rounding to float ----
__device__ __forceinline__ float expf_libm(float x) {
    static const double T[32] = {
        0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0, 0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0,
        0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0, 0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
        0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0, 0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0,
        0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0, 0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
        0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0, 0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0,
        0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0 };
    if (!(x >= -0x1.9fe368p6f)) return x != x ? x : 0.0f;       // underflow (softmax arguments are <= 0); -inf -> 0
    if (x > 0x1.62e42ep6f) return INFINITY;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double z = InvLn2N * (double) x;
    double kd = z + SHIFT;
    const uint64_t ki = (uint64_t) __double_as_longlong(kd);
    kd -= SHIFT;
    const double r = z - kd;
    // s = 2^(k/32): table value with the integer part of k/32 added to the exponent
    const uint64_t t = ((uint64_t) __double_as_longlong(T[ki & 31]) - ((ki & 31) << 47)) + (ki << 47);
    const double s = __longlong_as_double((long long) t);
    const double zz = fma(C0, r, C1), r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    return (float) (y * s);
}

// one lane of ggml_v_expf, AVX2 variant (ggml-cpu/vec.h:1215-1252)
__device__ __forceinline__ float v_expf_lane(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = __float_as_uint(z) << 23;
    const float k = __uint_as_float(e + __float_as_uint(1.0f));
    const bool c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, 0x1.ffffecp-1f * b);
    if (!c) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = __uint_as_float(g + 0x7f000000u);
    const float s2 = __uint_as_float(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}

__device__ __forceinline__ float rl_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// [key][dim] f16 tile in LDS, 16-byte slot index XOR-swizzled by the key so that the row-per-lane stores spread over the banks
__device__ __forceinline__ int vt_off(int key, int d) { return key*64 + ((((d >> 3) ^ (key & 7)) << 3) | (d & 7)); }

__device__ __forceinline__ void unpack8(const uint4 r, float * f) {
    const uint32_t w[4] = { r.x, r.y, r.z, r.w };
    #pragma unroll
    for (int e = 0; e < 4; e++) { f[2*e] = h2f((uint16_t) (w[e] & 0xFFFF)); f[2*e + 1] = h2f((uint16_t) (w[e] >> 16)); }
}

// ---------------------------------------------------------------------------------------------------------------------
// vec path: grid (nchunks, H, T), one wave.  Lane j scores key j of a 64-key tile (AVX2 order), then the wave walks the
// tile's keys one by one: lane d owns dim d of the F16 accumulator.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fattn_exact_vec(const FXArgs a) {
    __shared__ float qs[64];
    __shared__ __attribute__((aligned(16))) uint16_t vt[64*64];
    const int lane = threadIdx.x, c = blockIdx.x, h = blockIdx.y, t = blockIdx.z;
    const int hk = h / a.rk2, hv = h / a.rv2;
    const int ic0 = c * a.chunk, ic1 = min(ic0 + a.chunk, a.n_kv);
    const float * qp = (const float *) (a.q.data + (int64_t) t*a.q.nb[1] + (int64_t) h*a.q.nb[2]);
    qs[lane] = round_f16(qp[lane]);                                     // q_to_vec_dot: f32 -> f16 (ops.cpp:8597)
    __syncthreads();
    half_t acc = (half_t) 0.0f;
    float S = 0.0f, M = -INFINITY;
    for (int base = ic0; base < ic1; base += 64) {
        const int n = min(64, ic1 - base);
        const bool live = lane < n;
        const int key = live ? base + lane : ic1 - 1;
        const uint4 * kp = (const uint4 *) (a.k.data + (int64_t) key*a.k.nb[1] + (int64_t) hk*a.k.nb[2]);
        const uint4 * vp = (const uint4 *) (a.v.data + (int64_t) key*a.v.nb[1] + (int64_t) hv*a.v.nb[2]);
        uint4 kr[8];
        #pragma unroll
        for (int i = 0; i < 8; i++) kr[i] = kp[i];
        const float mv = a.has_mask ? h2f(*(const uint16_t *) (a.m.data + (int64_t) t*a.m.nb[1] + (int64_t) key*2)) : 0.0f;
        __syncthreads();                                                 // the previous tile's walk is done with vt
        #pragma unroll
        for (int i = 0; i < 8; i++) { const uint4 vrow = vp[i]; *(uint4 *) &vt[vt_off(lane, i*8)] = vrow; }
        // ggml_vec_dot_f16: accumulator (j, l) takes elements j*8 + l of each 32-element step, fused multiply-add
        float s[4][8];
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            float kf[8];
            unpack8(kr[j], kf);
            #pragma unroll
            for (int l = 0; l < 8; l++) s[j][l] = fmaf(kf[l], qs[j*8 + l], 0.0f);
        }
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            float kf[8];
            unpack8(kr[4 + j], kf);
            #pragma unroll
            for (int l = 0; l < 8; l++) s[j][l] = fmaf(kf[l], qs[32 + j*8 + l], s[j][l]);
        }
        float vv[8];
        #pragma unroll
        for (int l = 0; l < 8; l++) { const float x02 = s[0][l] + s[2][l], x13 = s[1][l] + s[3][l]; vv[l] = x02 + x13; }
        const float t0 = vv[0] + vv[4], t1 = vv[1] + vv[5], t2 = vv[2] + vv[6], t3 = vv[3] + vv[7];
        float sc = (t0 + t1) + (t2 + t3);
        sc = sc * a.scale;
        sc = sc + mv;
        const int skip = (!live || mv == -INFINITY) ? 1 : 0;
        __syncthreads();
        for (int jj = 0; jj < n; jj++) {
            if (__builtin_amdgcn_readlane(skip, jj)) continue;
            const float sj = rl_f(sc, jj);
            float vs = 1.0f;
            if (sj > M) {                                                // new maximum: rescale what has been accumulated
                const float ms = expf_libm(M - sj);
                M = sj;
                acc = (half_t) ((float) acc * ms);                       // ggml_vec_scale_f16: f32 multiply, stored as f16
                S = S * ms;
            } else vs = expf_libm(sj - M);
            acc = (half_t) fmaf(h2f(vt[vt_off(jj, lane)]), vs, (float) acc);   // ggml_vec_mad_f16: f32 fma, stored as f16
            S = S + vs;
        }
    }
    if (a.nchunks > 1) {
        float * rec = a.part + ((int64_t) h*a.nchunks + c) * 66;
        if (lane == 0) { rec[0] = M; rec[1] = S; }
        rec[2 + lane] = (float) acc;
    } else {
        const float inv = S == 0.0f ? 0.0f : 1.0f / S;
        *(float *) (a.d.data + (int64_t) lane*4 + (int64_t) h*a.d.nb[1] + (int64_t) t*a.d.nb[2]) = (float) acc * inv;
    }
}

// ggml_flash_attn_ext_reduce_partials (ops.cpp:8992-9075): grid (H), lane = dim, chunks merged in order
__global__ void __launch_bounds__(64) k_fattn_exact_reduce(const FXArgs a) {
    const int lane = threadIdx.x, h = blockIdx.x;
    float Mf = -INFINITY, Sf = 0.0f, fin = 0.0f;
    for (int c = 0; c < a.nchunks; c++) {
        if ((int64_t) c * a.chunk >= a.n_kv) continue;
        const float * rec = a.part + ((int64_t) h*a.nchunks + c) * 66;
        const float Mc = rec[0], Sc = rec[1], pv = rec[2 + lane];
        if (Sc == 0.0f) continue;
        const float Mn = fmaxf(Mf, Mc);
        const float so = expf_libm(Mf - Mn), sn = expf_libm(Mc - Mn);
        fin = fmaf(fin, so, pv * sn);                                    // a*b + c*d as the reference build contracts it
        Sf  = fmaf(Sf, so, Sc * sn);
        Mf = Mn;
    }
    if (Sf != 0.0f) fin = fin * (1.0f / Sf);
    *(float *) (a.d.data + (int64_t) lane*4 + (int64_t) h*a.d.nb[1]) = fin;
}

// ---------------------------------------------------------------------------------------------------------------------
// tiled path, one query row per wave (the rows of a 64-query tile are independent): grid (H, T)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fattn_exact_tiled(const FXArgs a) {
    __shared__ float qs[64];
    __shared__ __attribute__((aligned(16))) uint16_t vt[64*64];
    const int lane = threadIdx.x, h = blockIdx.x, t = blockIdx.y;
    const int hk = h / a.rk2, hv = h / a.rv2;
    const float * qp = (const float *) (a.q.data + (int64_t) t*a.q.nb[1] + (int64_t) h*a.q.nb[2]);
    qs[lane] = qp[lane];                                                 // F32 query, not rounded
    __syncthreads();
    float vkq = 0.0f, S = 0.0f, M = -INFINITY;
    for (int base = 0; base < a.n_kv; base += 64) {
        const int n = min(64, a.n_kv - base);
        const bool live = lane < n;
        const int key = live ? base + lane : a.n_kv - 1;
        const uint4 * kp = (const uint4 *) (a.k.data + (int64_t) key*a.k.nb[1] + (int64_t) hk*a.k.nb[2]);
        const uint4 * vp = (const uint4 *) (a.v.data + (int64_t) key*a.v.nb[1] + (int64_t) hv*a.v.nb[2]);
        uint4 kr[8];
        #pragma unroll
        for (int i = 0; i < 8; i++) kr[i] = kp[i];
        const float mv = a.has_mask ? h2f(*(const uint16_t *) (a.m.data + (int64_t) t*a.m.nb[1] + (int64_t) key*2)) : 0.0f;
        __syncthreads();
        #pragma unroll
        for (int i = 0; i < 8; i++) { const uint4 vrow = vp[i]; *(uint4 *) &vt[vt_off(lane, i*8)] = vrow; }
        float sc = 0.0f;                                                 // simd_gemm: sequential fma over the head dimension
        #pragma unroll
        for (int i = 0; i < 8; i++) {
            float kf[8];
            unpack8(kr[i], kf);
            #pragma unroll
            for (int l = 0; l < 8; l++) sc = fmaf(kf[l], qs[i*8 + l], sc);
        }
        sc = sc * a.scale;
        if (a.has_mask) sc = sc + mv;
        if (!live) sc = -INFINITY;
        __syncthreads();
        const float tmax = wave_max(sc);
        if (tmax == -INFINITY) continue;
        const float Mnew = fmaxf(M, tmax);
        if (Mnew > M) { const float ms = expf_libm(M - Mnew); vkq = vkq * ms; S = S * ms; }
        M = Mnew;
        const float p = v_expf_lane(sc - Mnew);
        // ggml_vec_soft_max_f32 (vec.cpp:541-551): per 8 lanes (p0+p4, p1+p5, p2+p6, p3+p7) -> (.0+.2) + (.1+.3), summed in double
        const float t4 = p + __shfl_down(p, 4, 64);
        const float t2 = t4 + __shfl_down(t4, 2, 64);
        const float t1 = t2 + __shfl_down(t2, 1, 64);
        double sum = 0.0;
        #pragma unroll
        for (int g = 0; g < 8; g++) sum += (double) rl_f(t1, 8*g);
        S = (float) ((double) S + sum);
        for (int kk = 0; kk < n; kk++) vkq = fmaf(h2f(vt[vt_off(kk, lane)]), rl_f(p, kk), vkq);     // keys >= n carry p = 0
    }
    const float inv = S == 0.0f ? 0.0f : 1.0f / S;
    *(float *) (a.d.data + (int64_t) lane*4 + (int64_t) h*a.d.nb[1] + (int64_t) t*a.d.nb[2]) = vkq * inv;
}

extern "C" int mi355x_flash_attn_ext_exact(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                                           const mi355x_tensor * mask, const mi355x_tensor * dst, float scale, int nth) {
    if (q->type != MI355X_TYPE_F32 || k->type != MI355X_TYPE_F16 || v->type != MI355X_TYPE_F16 || dst->type != MI355X_TYPE_F32) return MI355X_E_UNSUPPORTED;
    if (q->ne[0] != 64 || k->ne[0] != 64 || v->ne[0] != 64 || dst->ne[0] != 64) return MI355X_E_UNSUPPORTED;
    if (q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return MI355X_E_UNSUPPORTED;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2 || dst->nb[0] != 4) return MI355X_E_UNSUPPORTED;
    const int T = (int) q->ne[1], H = (int) q->ne[2], n_kv = (int) k->ne[1];
    if (v->ne[1] != n_kv || dst->ne[1] != H || dst->ne[2] != T) return MI355X_E_UNSUPPORTED;
    if (k->ne[2] <= 0 || H % k->ne[2] || v->ne[2] <= 0 || H % v->ne[2]) return MI355X_E_UNSUPPORTED;
    if (((uintptr_t) k->data | k->nb[1] | k->nb[2]) % 16 || ((uintptr_t) v->data | v->nb[1] | v->nb[2]) % 16 || ((uintptr_t) q->data | q->nb[1] | q->nb[2]) % 4) return MI355X_E_UNSUPPORTED;
    if (mask && (mask->type != MI355X_TYPE_F16 || mask->ne[0] < n_kv || mask->ne[1] < T || mask->nb[0] != 2 || mask->ne[2] != 1 || mask->ne[3] != 1)) return MI355X_E_UNSUPPORTED;
    if (T == 0 || H == 0) return 0;
    if (n_kv == 0) return mi355x_memset(ctx, dst->data, 0, (size_t) dst->nb[3]*dst->ne[3]);
    if (nth < 1) nth = 1;
    FXArgs a; memset(&a, 0, sizeof(a));
    a.q = to_d(q); a.k = to_d(k); a.v = to_d(v); a.d = to_d(dst);
    if (mask) a.m = to_d(mask);
    a.has_mask = mask != nullptr; a.scale = scale; a.T = T; a.n_kv = n_kv; a.H = H;
    a.rk2 = (int) (H / k->ne[2]); a.rv2 = (int) (H / v->ne[2]);
    if (T >= 64) return emit(ctx, "fattn_exact_tiled", k_fattn_exact_tiled, dim3(H, T), dim3(64), 0, a, 0, 0);
    if (T == 1 && n_kv >= 512 && nth > 1) {                      // split-KV: the reference's result depends on its thread count
        a.chunk = (n_kv + nth - 1) / nth; a.nchunks = nth;
        mi355x_scratch_reset(ctx);
        a.part = (float *) mi355x_scratch_alloc(ctx, (size_t) H * nth * 66 * 4);
        if (!a.part) return (int) hipErrorOutOfMemory;
        const int rc = emit(ctx, "fattn_exact_vec", k_fattn_exact_vec, dim3(nth, H, 1), dim3(64), 0, a, 0, 0);
        if (rc) return rc;
        return emit(ctx, "fattn_exact_reduce", k_fattn_exact_reduce, dim3(H), dim3(64), 0, a, 0, 0);
    }
    // (one thread and T == 1: the reference still takes the split path, with a single chunk; its merge is fma(0, 0, x * 1) = x
    //  followed by the same multiplication by 1 / S: identical words to the direct path)
    a.chunk = n_kv; a.nchunks = 1;
    return emit(ctx, "fattn_exact_vec", k_fattn_exact_vec, dim3(1, H, T), dim3(64), 0, a, 0, 0);
}
