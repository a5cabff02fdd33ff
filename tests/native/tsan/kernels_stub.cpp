// TEST infrastructure (never shipped): stand-in for libmi355x_kernels.so in the ThreadSanitizer run of the plugin's host logic.
// The compute entry points of include/mi355x_kernels.h are in kernels_stub_ops.c (they succeed without doing anything); here are the few
// whose RESULTS the host logic uses.
#include "mi355x_kernels.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

struct mi355x_ctx { void * stream; void * scratch[2]; std::atomic<uint64_t> n_eager; char parts[1 << 16]; };

extern "C" {
MI355X_API int          mi355x_device_count(void) { return 2; }
MI355X_API mi355x_ctx * mi355x_ctx_create(int) {
    mi355x_ctx * c = new mi355x_ctx();
    c->stream = calloc(1, 16); c->scratch[0] = calloc(1, 512 << 10); c->scratch[1] = calloc(1, 512 << 10);
    return c;
}
MI355X_API void         mi355x_ctx_destroy(mi355x_ctx * c) { if (c) { free(c->stream); free(c->scratch[0]); free(c->scratch[1]); delete c; } }
MI355X_API void *       mi355x_ctx_stream(mi355x_ctx * c) { return c->stream; }
MI355X_API const char * mi355x_last_error(void) { return ""; }
MI355X_API uint64_t     mi355x_eager_count(mi355x_ctx * c) { return c->n_eager.load(); }
MI355X_API void         mi355x_prof_enable(mi355x_ctx *, int) { }
MI355X_API int          mi355x_prof_report(mi355x_ctx *, mi355x_prof_row *, int) { return 0; }
MI355X_API void         mi355x_prof_reset(mi355x_ctx *) { }
MI355X_API int          mi355x_last_launch_mirrored(mi355x_ctx *) { return 1; }          // the logits-mirror hand-off is part of what is being checked
MI355X_API void *       mi355x_act_scratch(mi355x_ctx * c, int which) { return c->scratch[which & 1]; }
MI355X_API size_t       mi355x_act_planes_bytes(int, int K, int T) { return (size_t) K * T * 2; }
MI355X_API size_t       mi355x_act_rows_bytes(int, int64_t K, int64_t T) { return (size_t) (K * T * 2); }
MI355X_API int          mi355x_type_is_quantized(int t) { return t == MI355X_TYPE_Q4_0 || t == MI355X_TYPE_Q5_0 || t == MI355X_TYPE_Q8_0 || t == MI355X_TYPE_Q4_K; }
MI355X_API size_t       mi355x_type_row_bytes(int t, int64_t ne0) {
    switch (t) {
        case MI355X_TYPE_F32: case MI355X_TYPE_I32: return (size_t) ne0 * 4;
        case MI355X_TYPE_F16:  return (size_t) ne0 * 2;
        case MI355X_TYPE_Q4_0: return (size_t) ne0 / 32 * 18;
        case MI355X_TYPE_Q5_0: return (size_t) ne0 / 32 * 22;
        case MI355X_TYPE_Q8_0: return (size_t) ne0 / 32 * 34;
        case MI355X_TYPE_Q4_K: return (size_t) ne0 / 256 * 144;
    }
    return 0;
}
MI355X_API int mi355x_repack_to_planar(int t, const void * src, void * dst, int64_t n)   { memcpy(dst, src, mi355x_type_row_bytes(t, n)); return 0; }
MI355X_API int mi355x_repack_from_planar(int t, const void * src, void * dst, int64_t n) { memcpy(dst, src, mi355x_type_row_bytes(t, n)); return 0; }
static int fill_parts(mi355x_ctx * c, int T, int H, mi355x_attn_partials * out) {
    out->part_o = (const float *) c->parts; out->part_ml = (const float *) c->parts; out->nparts = 1; out->T = T; out->H = H;
    return 0;
}
MI355X_API int mi355x_flash_attn_partial(mi355x_ctx * c, const mi355x_tensor * q, const mi355x_tensor *, const mi355x_tensor *, const mi355x_tensor *, float, mi355x_attn_partials * out) {
    return fill_parts(c, (int) q->ne[1], (int) q->ne[2], out);
}
MI355X_API int mi355x_flash_attn_partial_multi(mi355x_ctx * c, int S, const mi355x_attn_state *, const mi355x_tensor * q, const mi355x_tensor *, const mi355x_tensor *, float, mi355x_attn_partials * out) {
    return fill_parts(c, S, (int) q->ne[2], out);
}
MI355X_API void mi355x_gelu_table_host(uint16_t * out) { memset(out, 0, 65536 * 2); }
MI355X_API int  mi355x_log_mel_n_len(int n) { return (n + 480000) / 160; }
MI355X_API void mi355x_test_option(int, int, int) { }
MI355X_API int  mi355x_debug_read_stamps(mi355x_ctx *, unsigned long long *) { return MI355X_E_UNSUPPORTED; }
}
