/*
 * mi355x_kernels.h — C ABI of the MI355X (gfx950 / CDNA4) kernel library `libmi355x_kernels.so`.
 *
 * This is the "thin C-ABI" between the C++ host side of the ggml backend plugin (libggml-mi355x.so,
 * see ggml_mi355x.h) and the hand-written HIP kernels.  Plain pointers and sizes only: no ggml, torch
 * or C++ types cross this boundary, so the same entry points are bound from ctypes in tests/ and bench.py.
 *
 * Every op mirrors the semantics of one ggml op on the whisper.cpp hot path; the reference definition
 * each one replaces is cited per function (paths relative to the reference tree).
 *
 * Data layout contract
 * --------------------
 *  - F32 / F16 / I32 tensors: ggml layout (ne[] element counts, nb[] BYTE strides), any strides the
 *    reference CPU kernel accepts unless stated otherwise.
 *  - Block-quantized tensors (Q4_0, Q5_0, Q8_0, Q4_K) live in HBM in a PLANAR ("struct of arrays")
 *    re-arrangement of ggml's block structs (ggml/src/ggml-common.h:194-199, :229-235, :251-256, :327-338)
 *    so that a wavefront reads them with aligned 16-byte loads.  For a tensor with NB blocks, in row-major
 *    block order, the tensor's bytes [0, nbytes) hold:
 *        Q4_0 : qs[NB][16]               | d[NB] (f16)
 *        Q5_0 : qs[NB][16] | qh[NB] (u32) | d[NB] (f16)
 *        Q8_0 : qs[NB][32]               | d[NB] (f16)
 *        Q4_K : qs[NB][128] | scales[NB][12] | dm[NB] (f16 d, f16 dmin)
 *    The byte count is identical to ggml's (18/22/34/144 bytes per block); mi355x_repack_to_planar /
 *    mi355x_repack_from_planar convert on the host.  nb[] of a quantized tensor keeps ggml's values and
 *    is only used to derive row/batch INDICES; such tensors must be contiguous.
 *
 * Return convention: 0 = ok, MI355X_E_UNSUPPORTED (-1) = shape/type not handled (caller falls back),
 * any positive value = hipError_t from the runtime.  A missing GPU makes mi355x_ctx_create return NULL.
 */
#ifndef MI355X_KERNELS_H
#define MI355X_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_API __attribute__((visibility("default")))

#define MI355X_E_UNSUPPORTED (-1)

/* numeric values == enum ggml_type (ggml/include/ggml.h:389-433) */
enum mi355x_type {
    MI355X_TYPE_F32  = 0,
    MI355X_TYPE_F16  = 1,
    MI355X_TYPE_Q4_0 = 2,
    MI355X_TYPE_Q5_0 = 6,
    MI355X_TYPE_Q8_0 = 8,
    MI355X_TYPE_Q4_K = 12,
    MI355X_TYPE_I32  = 26,
};

/* mirrors the fields of struct ggml_tensor (ggml/include/ggml.h:673-705) a kernel needs */
typedef struct mi355x_tensor {
    void *  data;      /* device pointer */
    int32_t type;      /* enum mi355x_type */
    int32_t reserved;
    int64_t ne[4];     /* elements per dim */
    int64_t nb[4];     /* byte strides */
} mi355x_tensor;

typedef struct mi355x_ctx mi355x_ctx;

/* ---- context: one HIP stream + scratch arena + constant tables (GELU f16 LUT) ------------------- */
MI355X_API int          mi355x_device_count(void);                      /* gfx950 devices visible; 0 if none */
MI355X_API mi355x_ctx * mi355x_ctx_create(int device);                  /* NULL on failure */
MI355X_API void         mi355x_ctx_destroy(mi355x_ctx * ctx);
MI355X_API void *       mi355x_ctx_stream(mi355x_ctx * ctx);            /* hipStream_t */
MI355X_API int          mi355x_ctx_synchronize(mi355x_ctx * ctx);
/* Independent neighbouring GEMMs are held back so that they can leave as ONE grouped launch (the Q / K / V projections of an encoder
 * layer, the cross-attention K / V projections of consecutive layers: src/whisper.cpp:2119-2141, :2306-2348); every other launch,
 * synchronize, memset and profile call flushes them first, so stream order equals program order.  A host that enqueues
 * its OWN work on mi355x_ctx_stream() calls this before. */
MI355X_API int          mi355x_flush(mi355x_ctx * ctx);
MI355X_API const char * mi355x_last_error(void);

/* n small host-to-device copies in ONE launch on `stream`: src_dev[i] are DEVICE addresses of pinned, device-mapped host memory
 * (hipHostGetDevicePointer).  Used by the plugin for the per-step graph inputs (ggml-backend.cpp:1625-1632: token ids, positions, mask). */
#define MI355X_SCATTER_MAX 8
MI355X_API int mi355x_scatter_upload(void * hip_stream, int n, void * const * dst, const void * const * src_dev, const uint32_t * sizes);
MI355X_API uint64_t mi355x_eager_count(mi355x_ctx * ctx); /* launches issued on the stream since the context was created */

/* profiling: when enabled every eager launch is bracketed by hipEvents on the context's stream and
 * accumulated per kernel name.  mi355x_prof_report fills up to `cap` rows; returns the row count. */
typedef struct mi355x_prof_row {
    const char * name;
    uint64_t     calls;
    double       total_ms;
    double       algo_bytes;   /* summed */
    double       algo_flops;   /* summed */
} mi355x_prof_row;
MI355X_API void mi355x_prof_enable(mi355x_ctx * ctx, int on);
MI355X_API int  mi355x_prof_report(mi355x_ctx * ctx, mi355x_prof_row * rows, int cap);  /* syncs the stream */
MI355X_API void mi355x_prof_reset(mi355x_ctx * ctx);

/* ---- host-side weight re-layout (ggml block structs <-> planar), see header comment -------------- */
MI355X_API int    mi355x_type_is_quantized(int type);
MI355X_API size_t mi355x_type_row_bytes(int type, int64_t ne0);      /* == ggml_row_size */
MI355X_API int    mi355x_repack_to_planar  (int type, const void * ggml_blocks, void * planar, int64_t nelements);
MI355X_API int    mi355x_repack_from_planar(int type, const void * planar, void * ggml_blocks, int64_t nelements);

/* ---- fused epilogue of a mul_mat (each field optional) -------------------------------------------
 * Applied in the order the whisper graphs apply the separate ggml ops (src/whisper.cpp:2119-2235,
 * :2550-2827): dst = act( (W·x + bias) * scale ) + residual, every step rounded to f32 like the
 * separate ggml_add / ggml_scale / ggml_gelu / ggml_add nodes. */
typedef struct mi355x_epilogue {
    const float * bias;            /* [N] broadcast over columns (ggml_add with a [N] or [N,1] src1) */
    float         scale;           /* used when has_scale */
    int32_t       has_scale;
    int32_t       gelu;            /* 1: ggml_gelu (f16-table GELU, ggml-cpu/vec.h:987-1000) */
    int32_t       bias_per_col;    /* 1: bias is [T], one value per COLUMN of dst (ggml_add with a [1,T] src1: the conv bias of
                                    * src/whisper.cpp:2013-2020, whose mul_mat has the im2col rows as src0).  MFMA path (T > 8) only;
                                    * every other kernel answers MI355X_E_UNSUPPORTED.  (fills the struct's former padding) */
    const float * residual;        /* [N,T] f32, same strides as dst; NULL = none */
    int64_t       residual_nb1;    /* byte stride between columns of residual */
} mi355x_epilogue;

/* ---- ops ------------------------------------------------------------------------------------------ */

/* ggml_mul_mat (ggml/src/ggml.c:3278; CPU kernel ggml-cpu/ggml-cpu.c:1254-1452).
 * dst[n,t] = sum_k w[k,n] * x[k,t].  w: Q4_0/Q5_0/Q8_0/Q4_K (planar) or F16/F32; x: F32 (or F16);
 * dst: F32 (or F16 when dst->type == F16: fused ggml_cpy into an F16 KV cache).
 * Like the CPU path, x is first rounded to the weight type's vec_dot_type (Q8_0 / Q8_K blocks,
 * ggml-cpu/ggml-cpu.c:1322-1357, arch/x86/quants.c:302-398; F16 for F16 weights). */
MI355X_API int mi355x_mul_mat(mi355x_ctx * ctx, const mi355x_tensor * w, const mi355x_tensor * x,
                              const mi355x_tensor * dst, const mi355x_epilogue * ep /* nullable */);

/* Two-step form of mul_mat for T > 8 (lets the caller share one prepared activation between several
 * weights, e.g. Q/K/V): mi355x_prep_act rounds x [K,T] (F32 or F16, row stride x_nb1 bytes) to the weight
 * type's vec_dot_type and stores it as f16 [T][K]: mode 0 = F16 weights (plain f16), 1 = Q4_0/Q5_0/Q8_0
 * weights (Q8_0 round trip), 2 = Q4_K weights (Q8_K round trip).  mi355x_gemm_f16act is the MFMA GEMM.
 * Modes 3 / 4 write the activation ROWS of the int8 tile GEMM instead (Q8_0 / Q8_K blocks as integers + scales, see mi355x_gemm_q8act;
 * act_f16 then points to mi355x_act_rows_bytes(...) bytes; K % 128 == 0). */
MI355X_API int mi355x_prep_act(mi355x_ctx * ctx, const void * x, int64_t x_nb1, int x_is_f16, void * act_f16, int K, int64_t T, int mode);
MI355X_API int mi355x_gemm_f16act(mi355x_ctx * ctx, const mi355x_tensor * w, const void * act_f16, int64_t ld, int64_t T,
                                  void * dst, int64_t dst_nb1, int dst_type, const mi355x_epilogue * ep /* nullable */);
/* the same product, whose epilogue ALSO leaves mi355x_prep_act(mode 1) of the F32 result in prep_out (f16 [T][M], M % 32 == 0): the result
 * is the activation matrix of the next quantized-weight GEMM (fc1 + GELU -> fc2, src/whisper.cpp:2224-2238), bit-identical to a separate
 * mi355x_prep_act pass over dst.  dst may be NULL when nothing else reads the F32 result (then only prep_out is written). */
MI355X_API int mi355x_gemm_f16act_prep(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act_f16, int64_t ldb, int64_t T,
                                       void * dst, int64_t dst_nb1, const mi355x_epilogue * ep, void * prep_out);

/* ---- the int8 tile GEMM over the QUANTIZED operands (T > 8; csrc/kernels/mmq.hip) ---------------------------------------------
 * The reference quantizes src1 to the weight type's vec_dot_type and runs integer block dots for EVERY column count
 * (ggml-cpu/ggml-cpu.c:1322-1357; arch/x86/quants.c:1142-1181 ...).  Here the same integers go through v_mfma_i32_32x32x32_i8 (one
 * instruction = one 32-element block of a 32 x 32 tile) and each block's sums are folded into F32 with fma(float(sum), dw*dx, acc):
 * the CPU's own integer sums, f32 summation in ascending block order.  The weight is read in its block-quantized planar layout and
 * unpacked per workgroup into an LDS int8 tile; no f16 / int8 copy of a weight exists.
 *
 * Activation ROWS of x [K, T] (csrc/kernels/qrows.h): q[T][K] int8 | d[K/32][T] f32 for Q4_0 / Q5_0 / Q8_0 weights (Q8_0 blocks,
 * arch/x86/quants.c:302-398), q[T][K] int8 | d[K/256][T] f32 | bsum[K/32][T] i32 for Q4_K weights (Q8_K blocks, ggml-quants.c:2768-2805) —
 * the scale arrays are BLOCK-major (one block's scales of consecutive columns are contiguous);
 * mi355x_act_rows_bytes gives the size.  Producers: mi355x_prep_act / mi355x_norm_prep with mode 3 (Q8_0 rows) or 4 (Q8_K rows),
 * mi355x_flash_attn_ext_prep_rows, mi355x_gemm_q8act_prep (the epilogue of the GEMM that produces x).  K % 128 == 0.
 * mi355x_gemm_q8act_prep: the epilogue ALSO leaves the Q8_0 rows of the F32 result (K' = M, M % 128 == 0) in prep_rows_out — the
 * activations of the next quantized GEMM (fc1 + GELU -> fc2, src/whisper.cpp:2224-2238); dst may be NULL when nothing else reads the
 * F32 result.  Bit-identical to mi355x_prep_act(mode 3) on dst. */
MI355X_API size_t mi355x_act_rows_bytes(int wtype, int64_t K, int64_t T);
MI355X_API int mi355x_gemm_q8act(mi355x_ctx * ctx, const mi355x_tensor * w, const void * act_rows, int64_t T,
                                 void * dst, int64_t dst_nb1, int dst_type, const mi355x_epilogue * ep /* nullable */);
MI355X_API int mi355x_gemm_q8act_prep(mi355x_ctx * ctx, const mi355x_tensor * w, const void * act_rows, int64_t T,
                                      void * dst, int64_t dst_nb1, const mi355x_epilogue * ep, void * prep_rows_out);

/* One-time preparation of a quantized weight for the MFMA path: writes dst_f16[M][K] = f16(dequantized w), the
 * exact values mi355x_gemm_f16act feeds the matrix cores when given the quantized tensor.  A caller that keeps
 * this copy (HBM is plentiful) passes it to mi355x_gemm_f16act as an F16 tensor and still prepares the
 * activation with the ORIGINAL weight type's mode: results are bit-identical to the quantized call, the GEMM
 * inner loop just carries no dequantization.  */
MI355X_API int mi355x_dequant_f16(mi355x_ctx * ctx, const mi355x_tensor * w, void * dst_f16);

/* Decoder-step fusion (T <= 8 columns): [optional LayerNorm(x)*ln_w+ln_b] -> quantize -> up to 3
 * mat-vec products sharing x (Q,K,V) each with its own epilogue/destination.  Same arithmetic as the
 * unfused node sequence norm,mul,add,mul_mat,add,scale,cpy (src/whisper.cpp:2529-2598). */
typedef struct mi355x_gemv_seg {
    const void *  w;               /* planar quantized or f16 weight [K, N] */
    int32_t       wtype;
    int32_t       N;
    mi355x_epilogue ep;
    void *        dst;             /* [N, T] */
    int32_t       dst_type;        /* F32 or F16 */
    int32_t       reserved;
    int64_t       dst_nb1;         /* byte stride between columns */
} mi355x_gemv_seg;

typedef struct mi355x_gemv_desc {
    const float * x;               /* [K, T] f32 */
    int64_t       x_nb1;
    int32_t       K;
    int32_t       T;
    int32_t       has_norm;        /* 1: x <- norm(x, eps) * ln_w + ln_b  (ggml_norm, ggml-cpu/ops.cpp:3698-3765) */
    float         eps;
    const float * ln_w;
    const float * ln_b;
    int32_t       nseg;
    int32_t       reserved;
    mi355x_gemv_seg seg[3];
    /* alternative activation source (x == NULL): the partial records of a decode attention
     * (mi355x_flash_attn_partial below); the combine (ggml-cpu/ops.cpp:8479-8715 final normalisation) runs in the
     * mat-vec prologue, K must equal H*64.  Not combinable with has_norm. */
    const float * attn_part_o;
    const float * attn_part_ml;
    int32_t       attn_nparts;
    int32_t       reserved2;
    /* third activation source (x == NULL, no partials): activations ALREADY rounded to the weight type's vec_dot_type by
     * mi355x_act_prepare or by a producer's epilogue (planes_out below) — the T >= 3 / cross-state pipeline, see below */
    const void *  x_planes;
    /* optional: segment 0 (nseg == 1, N % 32 == 0, Q4_0/Q5_0/Q8_0 consumer) also leaves the Q8_0 planes of its result — the
     * activations of the mat-vec that consumes it (fc1 + GELU -> fc2, src/whisper.cpp:2797-2827) — laid out for K' = N and the same T;
     * planes_out_only: the F32 result itself is not stored (nothing else reads it) */
    void *        planes_out;
    int32_t       planes_out_only;
    int32_t       reserved3;
    /* optional per-column destinations / residuals (cross-state batches: column t belongs to another whisper_state); NULL: the
     * strided form dst + t*dst_nb1, residual + t*residual_nb1.  Honoured by the x_planes path only. */
    const struct mi355x_gemv_cols * cols;
} mi355x_gemv_desc;
MI355X_API int mi355x_gemv_fused(mi355x_ctx * ctx, const mi355x_gemv_desc * d);

/* ---- decoder steps with T = 3..8 columns (beam search: src/whisper.cpp:6486-6543 decodes n_tokens = beam size per step) and
 * cross-state batches (several whisper_states' single-token steps as the columns of ONE launch chain, weights read once) --------
 * The activation vector of a stage is rounded to the weight type's vec_dot_type ONCE (arch/x86/quants.c:302-398 Q8_0,
 * ggml-quants.c:2768-2805 Q8_K) into "planes" in HBM — byte for byte the image of the LDS planes the fused T <= 2 kernels build
 * in every workgroup — and every mat-vec workgroup only copies that image (Q8_0 family: lo[T][K/32] uint4 | hi[T][K/32] uint4 |
 * d[T][K/32] f32 | sum[T][K/32] i32; Q8_K: q[4][T][K/64] uint4 | d[T][K/256] f32 | sums[T][K/32] i32).  Same arithmetic, same
 * summation order per column as the fused kernels: a column's result does not depend on T or on which path computed it.
 * More than 8 columns (cross-state batches of up to 32 states) are carried as GROUPS of 8: the planes of T columns are ceil(T/8)
 * such images back to back, image g (columns 8g .. min(8g+8, T)-1) at byte offset g * mi355x_act_planes_bytes(wtype, K, 8); the
 * mat-vec kernels keep their weight rows in registers and pass over the images one after the other — the weights are still read
 * from HBM once per step, and a column's arithmetic is the one of the 8-column kernels (bit-identical for every T). */
#define MI355X_MAX_COLS 32
#define MI355X_IMG_COLS 8
typedef struct mi355x_gemv_cols {
    void *        dst[3][MI355X_MAX_COLS];     /* [segment][column] */
    const float * res[3][MI355X_MAX_COLS];     /* NULL entries where the segment has no residual */
    /* optional second F32 copy of segment 0's column t (single-segment launches of the vocabulary projection): e.g. pinned host memory
     * mapped into the device, so that the caller's read-back of the logits (src/whisper.cpp:2957-2963) needs no device-to-host copy.
     * Honoured with or without x_planes; dst / res above only with x_planes.  mi355x_last_launch_mirrored tells whether the launch that
     * mi355x_gemv_fused just issued wrote them (a kernel family without the option ignores them). */
    void *        mirror[MI355X_MAX_COLS];
} mi355x_gemv_cols;
MI355X_API int mi355x_last_launch_mirrored(mi355x_ctx * ctx);

typedef struct mi355x_act_desc {
    const float * x;               /* f32 [K, T], column stride x_nb1 bytes — or NULL with xcol / attention partials */
    int64_t       x_nb1;
    const float * xcol[MI355X_MAX_COLS];   /* xcol[0] != NULL: per-column vectors instead of x + t*x_nb1 */
    int32_t       K, T;
    int32_t       wtype;           /* weight type of the consumer: Q4_0/Q5_0/Q8_0 -> Q8_0 planes, Q4_K -> Q8_K planes */
    int32_t       has_norm;        /* ggml_norm + affine first (K <= 2048), as in mi355x_gemv_desc */
    float         eps;
    int32_t       reserved;
    const float * ln_w;
    const float * ln_b;
    const float * attn_part_o;     /* x == NULL && xcol[0] == NULL: combine of decode-attention partial records (K = H*64 <= 2048) */
    const float * attn_part_ml;
    int32_t       attn_nparts;
    int32_t       reserved2;
} mi355x_act_desc;
MI355X_API size_t mi355x_act_planes_bytes(int wtype, int K, int T);
MI355X_API int    mi355x_act_prepare(mi355x_ctx * ctx, const mi355x_act_desc * d, void * planes);
/* two plane buffers owned by the context (each holds any K <= 8192, T <= MI355X_MAX_COLS): a stage reads one while its epilogue fills the other */
MI355X_API void * mi355x_act_scratch(mi355x_ctx * ctx, int which);


/* Decode attention in two halves (T <= 8 queries, head_dim 64): mi355x_flash_attn_partial computes, per head, query and
 * 128-key chunk, the un-normalised partial (max score m, sum l, sum_k exp(s_k - m) v_k) of ggml_flash_attn_ext
 * (ggml/src/ggml.c:5418-5460; CPU ggml-cpu/ops.cpp:8479-8715) into context scratch memory that stays valid until the
 * next scratch-using op on the same context; mi355x_flash_attn_combine merges the chunks into dst F32 [64, H, T].
 * The backend normally skips the second call and hands the partials to mi355x_gemv_fused (O-projection). */
typedef struct mi355x_attn_partials {
    const float * part_o;          /* [H][T][nparts][64] */
    const float * part_ml;         /* [H][T][nparts][2]  */
    int32_t       nparts, T, H, reserved;
} mi355x_attn_partials;
MI355X_API int mi355x_flash_attn_partial(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                                         const mi355x_tensor * mask /* nullable */, float scale, mi355x_attn_partials * out);
MI355X_API int mi355x_flash_attn_combine(mi355x_ctx * ctx, const mi355x_attn_partials * p, const mi355x_tensor * dst);
/* decode attention for S single-query states in ONE launch (cross-state batch): state s has its own q [64, 1, H], K / V caches
 * (shapes and strides of k0 / v0, n_kv[s] keys) and mask row; records land as column s of a T = S partial set with
 * nparts = max_s ceil(n_kv[s] / 128) (chunks beyond a state's keys are empty records: weight 0 in the combine). */
typedef struct mi355x_attn_state { const void * q; const void * k; const void * v; const void * mask /* f16 row or NULL */; int32_t n_kv; int32_t reserved; } mi355x_attn_state;
MI355X_API int mi355x_flash_attn_partial_multi(mi355x_ctx * ctx, int S, const mi355x_attn_state * st, const mi355x_tensor * q0, const mi355x_tensor * k0,
                                               const mi355x_tensor * v0, float scale, mi355x_attn_partials * out);
/* decoder SELF-attention of T columns (at most 512 keys each) in one launch, ending in the Q8_0 activation planes of the output
 * projection (Q4_0 / Q5_0 / Q8_0 weights): the result of mi355x_flash_attn_partial[_multi] -> mi355x_act_prepare(partials), bit for bit,
 * without the record round trip and the second launch.  st[c]: column c's q [64, 1, H], K / V (shapes and strides of k0 / v0), mask row, keys. */
MI355X_API int mi355x_flash_attn_planes(mi355x_ctx * ctx, int T, const mi355x_attn_state * st, const mi355x_tensor * q0, const mi355x_tensor * k0,
                                        const mi355x_tensor * v0, float scale, void * planes);
/* head of S single-token decoder steps in ONE launch: dst[s] = te[tok[s]] + pe[pos[s]] (src/whisper.cpp:2524-2526) and the
 * F32 -> F16 cast of every state's mask row (ggml_cast, :2520); a state with tok == NULL / mask_f32 == NULL skips that half */
typedef struct mi355x_head_state { const int32_t * tok; const int32_t * pos; float * dst; const float * mask_f32; void * mask_f16; int32_t n_mask; int32_t reserved; } mi355x_head_state;
MI355X_API int mi355x_decode_head_multi(mi355x_ctx * ctx, int S, const mi355x_head_state * st, const mi355x_tensor * te, const mi355x_tensor * pe);
/* ggml_flash_attn_ext (ggml/src/ggml.c:5418-5460; CPU ggml-cpu/ops.cpp:8479-8715).
 * q: F32 [D, T, H] (any nb1/nb2), k/v: F16 [D, n_kv, H] views, mask: F16 [n_kv, >=T] or NULL,
 * dst: F32 [D, H, T].  D must be 64 (all Whisper models).  softmax(scale*q.k + mask) . v */
MI355X_API int mi355x_flash_attn_ext(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k,
                                     const mi355x_tensor * v, const mi355x_tensor * mask /* nullable */,
                                     const mi355x_tensor * dst, float scale);
/* T > 8: the same, also leaving mi355x_prep_act(mode 1) of dst seen as [T][H*64] in prep_out (f16) — the activation matrix of the
 * quantized-weight output projection that consumes it (src/whisper.cpp:2165-2167 -> :2199-2203); bit-identical to the separate pass */
MI355X_API int mi355x_flash_attn_ext_prep(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k,
                                          const mi355x_tensor * v, const mi355x_tensor * mask /* nullable */,
                                          const mi355x_tensor * dst, float scale, void * prep_out);

/* T > 8: the same, leaving the Q8_0 ROWS (mi355x_prep_act mode 3, see mi355x_gemm_q8act) of dst seen as [T][H*64] in rows_out — the
 * activations of the int8 tile GEMM that is the output projection; H*64 % 128 == 0 */
MI355X_API int mi355x_flash_attn_ext_prep_rows(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k,
                                               const mi355x_tensor * v, const mi355x_tensor * mask /* nullable */,
                                               const mi355x_tensor * dst, float scale, void * rows_out);

/* ggml_flash_attn_ext with the ARITHMETIC of the reference CPU dispatcher (ggml-cpu/ops.cpp:9077-9230), opt-in: split-KV over
 * `nth` chunks for T == 1 && n_kv >= 512 (the CPU's result depends on its thread count), the F32 tiled path with ggml_v_expf for
 * T >= 64, the sequential F16-accumulating vec path otherwise; scores in the AVX2 lane order, libm-identical expf.  Same
 * tensors as mi355x_flash_attn_ext.  Slow by construction (the key loop is sequential like the CPU's); used by the backend
 * under GGML_MI355X_EXACT=1 to compare free-running decodes token for token with the CPU reference. */
MI355X_API int mi355x_flash_attn_ext_exact(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k,
                                           const mi355x_tensor * v, const mi355x_tensor * mask /* nullable */,
                                           const mi355x_tensor * dst, float scale, int nth);

/* ggml_norm (ggml/src/ggml.c:3139; CPU ggml-cpu/ops.cpp:3698-3765) with optional fused affine
 * (the ggml_mul + ggml_add that always follow it in whisper, src/whisper.cpp:2109-2114). */
MI355X_API int mi355x_norm(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * dst, float eps,
                           const float * w /* nullable [ne0] */, const float * b /* nullable [ne0] */);
/* the same, also writing what mi355x_prep_act(dst) would write: the f16 [T][K] activation matrix (mode as in mi355x_prep_act) of the
 * MFMA GEMM that consumes this LayerNorm (encoder: src/whisper.cpp:2107-2115 -> :2119-2141, :2214-2222 -> :2224-2228), bit-identical
 * to the two-pass result.  x must be [K, T] with K <= 2048. */
MI355X_API int mi355x_norm_prep(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * dst, float eps,
                                const float * w, const float * b, void * prep_f16, int mode);

/* ggml_add / ggml_mul with broadcasting of src1 (CPU ggml-cpu/binary-ops.cpp:140-148). op: 0 add, 1 mul */
MI355X_API int mi355x_binary(mi355x_ctx * ctx, int op, const mi355x_tensor * a, const mi355x_tensor * b, const mi355x_tensor * dst);

/* ggml_scale: dst = x*s + b (CPU ggml-cpu/ops.cpp:4568-4620) */
MI355X_API int mi355x_scale(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * dst, float s, float b);

/* ggml_gelu (ggml/src/ggml.c:2781; CPU ggml_vec_gelu_f32, ggml-cpu/vec.h:987-1000: f16 lookup table) */
MI355X_API int mi355x_gelu(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * dst);
/* the element-wise ops of the voice-activity-detection graph (src/whisper.cpp:4545-4680: encoder ReLUs, LSTM gates, STFT magnitude):
 * ggml_relu / ggml_sigmoid / ggml_tanh / ggml_sqrt (CPU ggml-cpu/unary-ops.cpp:19-53), contiguous F32 */
enum mi355x_unary_op { MI355X_UNARY_RELU = 2, MI355X_UNARY_SIGMOID = 3, MI355X_UNARY_TANH = 4, MI355X_UNARY_SQRT = 5 };
MI355X_API int mi355x_unary(mi355x_ctx * ctx, int op, const mi355x_tensor * x, const mi355x_tensor * dst);
/* ggml_pad_reflect_1d (CPU ggml-cpu/ops.cpp:8149-8180): reflective padding of dim 0 by p0 / p1 elements, F32 (the VAD's STFT input) */
MI355X_API int mi355x_pad_reflect_1d(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * dst, int p0, int p1);

/* ggml_cpy / ggml_cont / ggml_dup / ggml_cast between F32 and F16 (CPU ggml-cpu/ops.cpp:17-654) */
MI355X_API int mi355x_cpy(mi355x_ctx * ctx, const mi355x_tensor * src, const mi355x_tensor * dst);

/* ggml_get_rows (ggml/src/ggml.c:3891; CPU ggml-cpu/ops.cpp:4850-5017): src F32/F16/quantized(planar), idx I32 */
MI355X_API int mi355x_get_rows(mi355x_ctx * ctx, const mi355x_tensor * src, const mi355x_tensor * idx, const mi355x_tensor * dst);
/* ggml_add(ggml_get_rows(src, idx), ggml_get_rows(add, add_idx)) in one launch: token + positional embedding of the
 * decoder (src/whisper.cpp:2524-2526).  add: F32 rows; idx, add_idx: 1-D I32 of equal length */
MI355X_API int mi355x_get_rows_add(mi355x_ctx * ctx, const mi355x_tensor * src, const mi355x_tensor * idx,
                                   const mi355x_tensor * add, const mi355x_tensor * add_idx, const mi355x_tensor * dst);

/* ggml_im2col, 1-D case used by ggml_conv_1d (ggml/src/ggml.c:4468-4565; CPU ggml-cpu/ops.cpp:6437-6517).
 * x: F32 [IW, IC, N], dst: F16 or F32 [IC*KW, OW, N] */
MI355X_API int mi355x_im2col_1d(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * dst,
                                int kw, int s0, int p0, int d0);

/* ggml_soft_max_ext (ggml/src/ggml.c:4093; CPU ggml-cpu/ops.cpp:5455-5565): rows of ne0, mask F32/F16 or NULL */
MI355X_API int mi355x_soft_max(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * mask /* nullable */,
                               const mi355x_tensor * dst, float scale, float max_bias);

/* ggml_rope_ext / ggml_rope_multi: modes NORMAL(0), NEOX(2), MROPE(8), VISION(24), IMROPE(40) incl. YaRN parameters (ggml/src/ggml.c:4168;
 * CPU ggml-cpu/ops.cpp:5822-6131; modes ggml/include/ggml.h:250-254).  pos: I32 [ne2] (x 4 for the multi-position modes: t | h | w | e).
 * Not on whisper's graph; standalone op. */
typedef struct mi355x_rope_params {
    int32_t n_dims, mode, n_ctx_orig;
    float   freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
    int32_t sections[4];           /* modes MROPE (8), VISION (24), IMROPE (40): dims per position stream t / h / w / e (ggml_rope_multi) */
} mi355x_rope_params;
MI355X_API int mi355x_rope(mi355x_ctx * ctx, const mi355x_tensor * x, const mi355x_tensor * pos /* I32 [ne2] */,
                           const float * freq_factors /* nullable */, const mi355x_tensor * dst,
                           const mi355x_rope_params * p);

/* arg-max of one F32 row of n values in HBM plus the runner-up: out16 (device-visible, 16 bytes) = { i32 index of the first maximum, f32 its
 * value, f32 the second largest value, i32 n } — device-side greedy sampling over the logits the decoder left in HBM (the reference's sampler
 * scans the row on the host, src/whisper.cpp:6486-6543) */
MI355X_API int mi355x_argmax_top2(mi355x_ctx * ctx, const float * x_dev, int n, void * out16);

/* ggml_concat along `dim` for F32 (CPU ggml-cpu/ops.cpp concat; dtw timestamps only, src/whisper.cpp:2741) */
MI355X_API int mi355x_concat(mi355x_ctx * ctx, const mi355x_tensor * a, const mi355x_tensor * b, const mi355x_tensor * dst, int dim);

/* the 65536-entry f16 GELU lookup table the kernels use (== ggml_table_gelu_f16, ggml-cpu/ggml-cpu.c:3847),
 * computed on the host; exposed so that tests can compare it with the reference's table without a GPU */
MI355X_API void mi355x_gelu_table_host(uint16_t * out65536);

/* TEST hook: kernel-shape variants that a test compares with the default inside one process (tests/test_gpu_mmq.py, tests/test_gpu_encoder.py).
 * Production never calls it; unset options take the built-in default.  set = 0 returns an option to its default. */
enum mi355x_test_opt {
    MI355X_OPT_MMQ_GROUP = 0,      /* 0: the int8 tile GEMM's products leave as single launches instead of grouped ones */
    MI355X_OPT_MMQ_SCALE_MFMA,     /* 0: scale products dw * dx on the VALU instead of the rank-1 f16 MFMA */
    MI355X_OPT_MMQ_TILE,           /* 12864: 128 x 64 tiles instead of 64 x 128 */
    MI355X_OPT_FATTN_NG,           /* 1..4: key groups per workgroup of the MFMA attention kernel instead of the n_kv-dependent choice */
    MI355X_OPT_DQ_GEMM,            /* 1: quantized weight x f16 activations on k_gemm_dq (planes unpacked per workgroup into LDS) instead of the register-staged k_gemm_mfma (bit-identical; the plugin sets it for GGML_MI355X_MMQ=3) */
    MI355X_OPT_DQ_BN,              /* 128 / 256: token-tile width of k_gemm_dq instead of the tile-count-dependent choice (bit-identical) */
    MI355X_OPT_DQ_ABLATE,          /* timing experiments on k_gemm_dq (results are garbage): bit 0 no unpack, 1 no activation DMA, 2 no MFMA, 3 no weight loads */
    MI355X_OPT_COUNT
};
MI355X_API void mi355x_test_option(int opt, int value, int set);

/* debug: with GGML_MI355X_KTIME=1 the decode mat-vec stamps s_memtime at its phase boundaries (workgroup 0, wave 0);
 * copies the 16 stamps of the last launch to the host.  MI355X_E_UNSUPPORTED when the mode is off. */
MI355X_API int mi355x_debug_read_stamps(mi355x_ctx * ctx, unsigned long long * out16);

/* wide empty launch on `stream` (hipStream_t): wakes the whole chip ahead of the first real dispatch of a decode step */
MI355X_API int mi355x_wake(void * stream, int nblocks);

/* whisper's log-mel front end (log_mel_spectrogram, src/whisper.cpp:3046-3283; whisper_pcm_to_mel :3901) for PCM resident in HBM:
 * pcm_dev f32 [n_samples] (16 kHz, [-1, 1]), filters_dev f32 [n_mel][201] (the model file's filterbank, whisper_filters), mel_dev f32
 * [n_mel][n_len] with n_len = mi355x_log_mel_n_len(n_samples) = (n_samples + 30 s) / 160 — exactly the layout whisper_set_mel takes
 * (data[j * n_len + i], src/whisper.cpp:2406).  Reflective 200-sample padding in front, 30 s of zeros behind, Hann 400 / hop 160,
 * power spectrum, filterbank (f64 sums), log10, clamp to (max - 8), (x + 4) / 4.  Runs on the context's stream. */
MI355X_API int mi355x_log_mel_n_len(int n_samples);
MI355X_API int mi355x_log_mel(mi355x_ctx * ctx, const float * pcm_dev, int n_samples, const float * filters_dev, int n_mel, int n_fft_bins /* 201 */,
                              float * mel_dev, int n_len);

/* order-independent 128-bit checksum of a device range on `stream` (hipStream_t): out[0] = sum of the 64-bit words, out[1] = sum of
 * word * (2*index + 1), mod 2^64; dev_out16 is 16 bytes of DEVICE memory.  Used to verify that every replica's weight buffers equal
 * rank 0's after the one-time broadcast. */
MI355X_API int mi355x_checksum(void * stream, const void * dptr, size_t nbytes, void * dev_out16);

/* memset / memcpy helpers on the context stream */
MI355X_API int mi355x_memset(mi355x_ctx * ctx, void * dptr, int value, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_KERNELS_H */
