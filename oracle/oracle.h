/*
 * oracle.h — CPU restatement of the reference arithmetic on the whisper.cpp hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the product path
 * (whisper.cpp_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * only as the checker.
 *
 * Every function restates (in plain scalar C, written for this repository — not copied) what the reference's
 * AVX2 CPU path computes, and cites the reference file:line it follows (paths relative to the reference tree,
 * ggml 0.21.0 as vendored by whisper.cpp v1.9.3).  Pinning: tests/test_oracle.py checks each function
 *   (a) against the reference's own compiled code in oracle/_ref (libggml-base.so / libggml-cpu.so) when present,
 *   (b) against committed golden vectors tests/golden/*.npz that were generated from oracle/_ref by
 *       tests/golden/make_golden.py.
 * The reference holds NO known-answer vectors of its own for the quantized mat-mul path, so the
 * pin is "executable reference + vectors generated from it".
 */
#ifndef WHISPER_ORACLE_H
#define WHISPER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml_type values used here (ggml/include/ggml.h:389-433) */
enum { ORACLE_F32 = 0, ORACLE_F16 = 1, ORACLE_Q4_0 = 2, ORACLE_Q5_0 = 6, ORACLE_Q8_0 = 8, ORACLE_Q4_K = 12 };

/* IEEE binary16 <-> binary32, round-to-nearest-even (what F16C does: ggml-impl.h GGML_COMPUTE_FP32_TO_FP16) */
uint16_t oracle_f32_to_f16(float f);
float    oracle_f16_to_f32(uint16_t h);

size_t oracle_row_size(int type, int64_t n);                 /* ggml_row_size */

/* block formats: ggml/src/ggml-common.h:194-199 (q4_0), :229-235 (q5_0), :251-256 (q8_0), :327-338 (q4_K) */
void oracle_dequantize_row(int type, const void * blocks, float * y, int64_t n);     /* ggml-quants.c:459,500,553,1529 */
void oracle_quantize_row_ref(int type, const float * x, void * blocks, int64_t n);   /* ggml-quants.c:113,187,276,1457 (weights) */

/* activation quantizers of the CPU mat-mul: Q8_0 as the AVX2 path computes it (round-to-nearest-EVEN,
 * ggml-cpu/arch/x86/quants.c:302-398) and Q8_K (ggml-quants.c:2768-2805) */
void oracle_quantize_row_q8_0(const float * x, void * blocks /* 34 B per 32 */, int64_t n);
void oracle_quantize_row_q8_K(const float * x, void * blocks /* 292 B per 256 */, int64_t n);

/* vec_dot of one weight row with one quantized activation row (ggml-cpu/quants.c:225-259 q4_0, :365-406 q5_0,
 * :451-479 q8_0, :696-769 q4_K; AVX2 kernels arch/x86/quants.c:701, :1142, :1308, :2038): integer dot per block,
 * f32 fma with d_w*d_x */
float oracle_vec_dot(int type, int64_t n, const void * w_blocks, const void * act_blocks);

/* ggml_mul_mat, 2-D: dst[n + t*N] = sum_k w[k,n] x[k,t]  (ggml-cpu/ggml-cpu.c:1254-1452: src1 -> vec_dot_type, then vec_dot).
 * w: `type` blocks (F16: halves, F32: floats), x: f32 [T][K], dst: f32 [T][N] */
void oracle_mul_mat(int type, const void * w, const float * x, float * dst, int64_t K, int64_t N, int64_t T);

void oracle_norm(const float * x, float * y, int64_t n, int64_t nrows, float eps);             /* ggml-cpu/ops.cpp:3698-3765 */
void oracle_gelu(const float * x, float * y, int64_t n);                                       /* ggml-cpu/vec.h:987-1000 (f16 LUT) */
void oracle_soft_max(const float * x, const float * mask, float * y, int64_t n, int64_t nrows, float scale); /* ops.cpp:5455-5565 */
void oracle_im2col_1d_f16(const float * x, uint16_t * dst, int64_t IW, int64_t IC, int64_t OW, int KW, int s0, int p0, int d0); /* ops.cpp:6437-6517 */
/* rope NORMAL(0)/NEOX(2), f32 (ggml-cpu/ops.cpp:5822-6131) ; x,y: [n_pos][n_head][ne0] */
void oracle_rope(const float * x, const int32_t * pos, float * y, int64_t ne0, int64_t n_head, int64_t n_pos, int n_dims, int mode,
                 int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow);
/* flash_attn_ext reference path (ggml-cpu/ops.cpp:8479-8715, `one_chunk`: q -> f16, f32 scores, online softmax,
 * V accumulated in F16).  q: f32 [T][H][D], k,v: f16 [n_kv][H][D], mask: f16 [T][n_kv] or NULL, dst: f32 [T][H][D] */
void oracle_flash_attn(const float * q, const uint16_t * k, const uint16_t * v, const uint16_t * mask, float * dst,
                       int64_t D, int64_t T, int64_t H, int64_t n_kv, float scale);

/* flash_attn_ext exactly as the AVX2 CPU build dispatches it (ggml-cpu/ops.cpp:9077-9230): split-KV over `nth` threads for
 * T == 1 && n_kv >= 512, the F32 tiled path for T >= 64, the F16-accumulating vec path otherwise.  Same layouts as above.
 * Bit-identical to oracle/_ref (tests/test_oracle.py).  oracle_v_expf: one lane of ggml_v_expf (vec.h:1215-1252). */
void  oracle_flash_attn_ext(const float * q, const uint16_t * k, const uint16_t * v, const uint16_t * mask, float * dst,
                            int64_t D, int64_t T, int64_t H, int64_t n_kv, float scale, int nth);
float oracle_v_expf(float x);

/* whisper's log-mel front end (log_mel_spectrogram, src/whisper.cpp:3046-3283) incl. its recursive FFT; pcm f32 [n] at 16 kHz,
 * filters f32 [n_mel][201], out f32 [n_mel][n_len], n_len = oracle_log_mel_n_len(n).  Pinned against the reference's own
 * front end run on samples/jfk.wav (tests/golden/mel.npz, made by tests/golden/make_golden.py through oracle/_ref/mel_ref). */
int64_t oracle_log_mel_n_len(int64_t n_samples);
void    oracle_log_mel(const float * pcm, int64_t n, const float * filters, int64_t n_mel, float * out);

#ifdef __cplusplus
}
#endif
#endif
