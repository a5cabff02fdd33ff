// Quantized activation ROWS — the wide (T > 32 columns) form of the reference's vec_dot_type blocks.
//
// ggml-cpu quantizes src1 of every quantized mul_mat to the weight type's vec_dot_type before its integer block dots
// (ggml-cpu/ggml-cpu.c:1322-1357): Q8_0 blocks for Q4_0 / Q5_0 / Q8_0 weights (arch/x86/quants.c:302-398: d = amax / 127 stored as
// f16, q = round(x / d)), Q8_K blocks for Q4_K weights (ggml-quants.c:2768-2805: one f32 scale per 256, sums per 16).  The int8 tile
// GEMM (mmq.hip) consumes exactly those integers.  For x [K, T] the bytes are, K-contiguous per column so that a 128-element K-step
// of a column is one 128-byte line:
//     Q8_0 rows : q[T][K] int8 | d[K/32][T]  f32 (f16-valued)
//     Q8_K rows : q[T][K] int8 | d[K/256][T] f32 | bsum[K/32][T] i32 (sum of the 32 q of a sub-block: pairs of the reference's bsums)
// The scales are BLOCK-major (one block's scales of consecutive columns are contiguous): a tile of the GEMM needs 4 blocks x 128
// columns of them per K-step, i.e. four 512-byte runs instead of 128 four-float snippets 4*K/32 bytes apart.
// (The decoder's "planes" — decode_common.h — are the same integers arranged for the mat-vec kernels' LDS image.)
#pragma once
#include <stdint.h>
#include <stddef.h>

#define MI355X_PREP_Q8_0_ROWS 3      // mi355x_prep_act / mi355x_norm_prep / ... `mode` values that write rows instead of f16 d*q
#define MI355X_PREP_Q8_K_ROWS 4

struct qrows_t { int8_t * q; float * d; int * bsum; };

static inline __host__ __device__ size_t qrows_bytes(int q8k, int64_t K, int64_t T) {
    return (size_t) T * (size_t) K + (size_t) T * (size_t) (q8k ? K / 256 : K / 32) * 4 + (q8k ? (size_t) T * (size_t) (K / 32) * 4 : 0);
}
static inline __host__ __device__ qrows_t qrows_of(void * base, int q8k, int64_t K, int64_t T) {
    qrows_t r;
    r.q = (int8_t *) base;
    r.d = (float *) ((char *) base + (size_t) T * (size_t) K);
    r.bsum = q8k ? (int *) (r.d + (size_t) T * (size_t) (K / 256)) : nullptr;
    return r;
}
