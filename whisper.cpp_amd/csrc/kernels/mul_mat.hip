// ggml_mul_mat dispatch (ggml/src/ggml.c:3278; CPU ggml-cpu/ggml-cpu.c:1254-1452):
//   T <= 8 columns            -> k_gemv   (gemv.hip, int8 dot, HBM-bound)
//   2-D, T > 8, quantized     -> k_prep_act (rows) + k_mmq (mmq.hip: int8 MFMA over the quantized operands)
//   2-D, T > 8, F16 / F32     -> k_prep_act + k_gemm_mfma / k_gemm_f16_ring (gemm_mfma.hip, f16 MFMA)
//   batched / strided / F32 A -> k_mul_mat_generic below (one wave per output element; only reached by the
//                                -nfa attention path and by f32 models, never by the default whisper graphs)
#include "common.h"


struct MMGenArgs { dtensor a, b, d; int at, bt; int64_t nbt; int64_t nout; int r2, r3; };

template <int AT>
__global__ void __launch_bounds__(256) k_mul_mat_generic(const MMGenArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= a.nout) return;
    int64_t r = o;
    const int64_t n = r % a.d.ne[0]; r /= a.d.ne[0];
    const int64_t t = r % a.d.ne[1]; r /= a.d.ne[1];
    const int64_t i2 = r % a.d.ne[2]; const int64_t i3 = r / a.d.ne[2];
    const int64_t K = a.a.ne[0];
    const char * bp = a.b.data + t*a.b.nb[1] + i2*a.b.nb[2] + i3*a.b.nb[3];
    float acc = 0.0f;
    if constexpr (AT == MI355X_TYPE_F32 || AT == MI355X_TYPE_F16) {
        const char * ap = a.a.data + n*a.a.nb[1] + (i2 / a.r2)*a.a.nb[2] + (i3 / a.r3)*a.a.nb[3];
        for (int64_t k = lane; k < K; k += 64) {
            // (x may be strided along k: the voice-activity LSTM feeds a transposed view, src/whisper.cpp:4598-4602)
            float x = a.bt == MI355X_TYPE_F32 ? *(const float *) (bp + k*a.b.nb[0]) : h2f(*(const uint16_t *) (bp + k*a.b.nb[0]));
            float w;
            if (AT == MI355X_TYPE_F16) { w = h2f(((const uint16_t *) ap)[k]); x = round_f16(x); }
            else w = ((const float *) ap)[k];
            acc = fmaf(w, x, acc);
        }
    } else {
        const qplanes<AT> p(a.a.data, a.nbt);
        const int64_t nb32 = K / 32;
        const int64_t rowidx = n + a.a.ne[1] * ((i2 / a.r2) + a.a.ne[2] * (i3 / a.r3));
        for (int64_t g = lane; g < nb32; g += 64) {
            float w[32];
            dequant_block32<AT>(p, rowidx*nb32 + g, w);
            #pragma unroll
            for (int j = 0; j < 32; j++) {
                const float x = a.bt == MI355X_TYPE_F32 ? *(const float *) (bp + (g*32 + j)*a.b.nb[0]) : h2f(*(const uint16_t *) (bp + (g*32 + j)*a.b.nb[0]));
                acc = fmaf(w[j], x, acc);
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) *(float *) (a.d.data + n*a.d.nb[0] + t*a.d.nb[1] + i2*a.d.nb[2] + i3*a.d.nb[3]) = acc;
}

static int mul_mat_generic(mi355x_ctx * ctx, const mi355x_tensor * w, const mi355x_tensor * x, const mi355x_tensor * dst) {
    if (dst->type != MI355X_TYPE_F32) return MI355X_E_UNSUPPORTED;
    if (x->type != MI355X_TYPE_F32 && x->type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
    if (x->nb[0] <= 0 || x->nb[0] % (x->type == MI355X_TYPE_F32 ? 4 : 2)) return MI355X_E_UNSUPPORTED;
    MMGenArgs k; k.a = to_d(w); k.b = to_d(x); k.d = to_d(dst); k.at = w->type; k.bt = x->type; k.nbt = 0;
    k.nout = t_nelements(dst);
    k.r2 = (int) (x->ne[2] / w->ne[2]); k.r3 = (int) (x->ne[3] / w->ne[3]);
    if (k.nout == 0) return 0;
    const int64_t nblk = (k.nout + 3) / 4;
    if (nblk > 0x7fffffffLL) return MI355X_E_UNSUPPORTED;
    const dim3 g((uint32_t) nblk), b(256);
    const double flops = 2.0 * (double) k.nout * w->ne[0];
    if (mi355x_type_is_quantized(w->type)) {
        if (!t_is_contiguous(w) || w->ne[0] % 32) return MI355X_E_UNSUPPORTED;
        k.nbt = t_nelements(w) / type_block(w->type);
    } else if (w->nb[0] != (w->type == MI355X_TYPE_F32 ? 4 : 2)) return MI355X_E_UNSUPPORTED;
    switch (w->type) {
        case MI355X_TYPE_F32:  return emit(ctx, "mul_mat_generic", k_mul_mat_generic<MI355X_TYPE_F32>,  g, b, 0, k, 0, flops);
        case MI355X_TYPE_F16:  return emit(ctx, "mul_mat_generic", k_mul_mat_generic<MI355X_TYPE_F16>,  g, b, 0, k, 0, flops);
        case MI355X_TYPE_Q4_0: return emit(ctx, "mul_mat_generic", k_mul_mat_generic<MI355X_TYPE_Q4_0>, g, b, 0, k, 0, flops);
        case MI355X_TYPE_Q5_0: return emit(ctx, "mul_mat_generic", k_mul_mat_generic<MI355X_TYPE_Q5_0>, g, b, 0, k, 0, flops);
        case MI355X_TYPE_Q8_0: return emit(ctx, "mul_mat_generic", k_mul_mat_generic<MI355X_TYPE_Q8_0>, g, b, 0, k, 0, flops);
        case MI355X_TYPE_Q4_K: return emit(ctx, "mul_mat_generic", k_mul_mat_generic<MI355X_TYPE_Q4_K>, g, b, 0, k, 0, flops);
        default: return MI355X_E_UNSUPPORTED;
    }
}

extern "C" int mi355x_mul_mat(mi355x_ctx * ctx, const mi355x_tensor * w, const mi355x_tensor * x, const mi355x_tensor * dst, const mi355x_epilogue * ep) {
    const int64_t K = w->ne[0], N = w->ne[1];
    if (x->ne[0] != K || dst->ne[0] != N || dst->ne[1] != x->ne[1] || dst->ne[2] != x->ne[2] || dst->ne[3] != x->ne[3]) return MI355X_E_UNSUPPORTED;
    if (w->ne[2] <= 0 || w->ne[3] <= 0 || x->ne[2] % w->ne[2] || x->ne[3] % w->ne[3]) return MI355X_E_UNSUPPORTED;
    if (t_nelements(dst) == 0) return 0;
    const bool two_d = w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1;
    const int64_t T = x->ne[1];
    const bool dst_ok = dst->nb[0] == (dst->type == MI355X_TYPE_F32 ? 4 : 2) && (dst->type == MI355X_TYPE_F32 || dst->type == MI355X_TYPE_F16);
    const bool wq = mi355x_type_is_quantized(w->type);
    const bool w_fast = (wq && t_is_contiguous(w)) || (w->type == MI355X_TYPE_F16 && w->nb[0] == 2 && w->nb[1] == K*2);

    if (two_d && dst_ok && w_fast && T <= 8 && x->type == MI355X_TYPE_F32 && x->nb[0] == 4) {
        mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
        d.x = (const float *) x->data; d.x_nb1 = x->nb[1]; d.K = (int) K; d.T = (int) T; d.nseg = 1;
        d.seg[0].w = w->data; d.seg[0].wtype = w->type; d.seg[0].N = (int) N;
        if (ep) d.seg[0].ep = *ep;
        d.seg[0].dst = dst->data; d.seg[0].dst_type = dst->type; d.seg[0].dst_nb1 = dst->nb[1];
        const int rc = mi355x_gemv_fused(ctx, &d);
        if (rc != MI355X_E_UNSUPPORTED) return rc;
    }
    if (two_d && dst_ok && K % 8 == 0 && (w_fast || (w->type == MI355X_TYPE_F16 && w->nb[0] == 2 && w->nb[1] % 16 == 0)) &&
        ((x->type == MI355X_TYPE_F32 && x->nb[0] == 4 && x->nb[1] % 16 == 0) || (x->type == MI355X_TYPE_F16 && x->nb[0] == 2 && x->nb[1] % 16 == 0)) &&
        ((uintptr_t) x->data % 16 == 0)) {
        const int mode = w->type == MI355X_TYPE_Q4_K ? 2 : (wq ? 1 : 0);
        // quantized weight, K in whole 128-element steps: the int8 tile GEMM on the reference's own Q8_0 / Q8_K integers (mmq.hip)
        static const bool mmq_on = !(getenv("GGML_MI355X_MMQ") && !atoi(getenv("GGML_MI355X_MMQ")));
        if (mmq_on && wq && t_is_contiguous(w) && K % 128 == 0 && (mode != 2 || K % 256 == 0) && T <= 65535) {
            mi355x_scratch_reset(ctx);
            void * rows = mi355x_scratch_alloc(ctx, mi355x_act_rows_bytes(w->type, K, T));
            if (!rows) return (int) hipErrorOutOfMemory;
            int rc = mi355x_prep_act(ctx, x->data, x->nb[1], x->type == MI355X_TYPE_F16, rows, (int) K, T, mode == 2 ? 4 : 3);
            if (rc == 0) rc = mi355x_gemm_q8act(ctx, w, rows, T, dst->data, dst->nb[1], dst->type, ep);
            if (rc != MI355X_E_UNSUPPORTED) return rc;
        }
        if (!(mode == 1 && K % 32) && !(mode == 2 && K % 256)) {
            const uint16_t * B; int64_t ldb;
            if (x->type == MI355X_TYPE_F16 && mode == 0) { B = (const uint16_t *) x->data; ldb = x->nb[1] / 2; }
            else {
                mi355x_scratch_reset(ctx);
                uint16_t * y = (uint16_t *) mi355x_scratch_alloc(ctx, (size_t) T*K*2);
                if (!y) return (int) hipErrorOutOfMemory;
                const int rc = mi355x_prep_act(ctx, x->data, x->nb[1], x->type == MI355X_TYPE_F16, y, (int) K, T, mode);
                if (rc) return rc;
                B = y; ldb = K;
            }
            const int rc = mi355x_gemm_f16act(ctx, w, B, ldb, T, dst->data, dst->nb[1], dst->type, ep);
            if (rc != MI355X_E_UNSUPPORTED) return rc;
        }
    }
    if (ep && (ep->bias || ep->has_scale || ep->gelu || ep->residual)) return MI355X_E_UNSUPPORTED;
    return mul_mat_generic(ctx, w, x, dst);
}
