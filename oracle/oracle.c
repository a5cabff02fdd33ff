/*
 * oracle.c — plain-C restatement of the whisper.cpp hot-path arithmetic (see oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, imported by or executed from the product path.
 * "parity pin": validated against the reference's compiled code in oracle/_ref and the golden vectors generated
 * from it (tests/test_oracle.py, tests/golden/make_golden.py).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- f16 ------------------------------------------------------------------------------------------ */
float oracle_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {                                   /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400));
            bits = sign | ((uint32_t) (127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

uint16_t oracle_f32_to_f16(float f) {            /* round-to-nearest-even, like F16C / _cvtss_sh(.., 0) */
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t) (sign | 0x7C00 | (x > 0x7F800000u ? 0x200 | ((x >> 13) & 0x3FF) : 0));
    if (x >= 0x477FF000u) return (uint16_t) (sign | 0x7C00);             /* >= 65520 rounds to inf */
    if (x < 0x33000001u) return sign;                                     /* <= 2^-25 rounds to zero */
    int32_t e = (int32_t) (x >> 23) - 127;
    uint32_t m = (x & 0x7FFFFF) | 0x800000;
    if (e < -14) {                                                        /* subnormal result */
        const int shift = -14 - e + 13;                                   /* 14..24 */
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t) (sign | r);
    }
    uint32_t r = ((uint32_t) (e + 15) << 10) | ((m >> 13) & 0x3FF);
    const uint32_t rem = m & 0x1FFF;
    if (rem > 0x1000 || (rem == 0x1000 && (r & 1))) r++;                  /* carry may bump the exponent: still correct */
    return (uint16_t) (sign | r);
}

/* ---- block layouts (ggml/src/ggml-common.h) --------------------------------------------------------- */
#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; } blk_q4_0;                    /* :194-199 */
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; } blk_q5_0;     /* :229-235 */
typedef struct { uint16_t d; int8_t qs[32]; } blk_q8_0;                     /* :251-256 */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; } blk_q4_K;  /* :327-338 */
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; } blk_q8_K;    /* :371-376 */
#pragma pack(pop)

size_t oracle_row_size(int type, int64_t n) {
    switch (type) {
        case ORACLE_F32: return (size_t) n * 4;
        case ORACLE_F16: return (size_t) n * 2;
        case ORACLE_Q4_0: return (size_t) (n / 32) * sizeof(blk_q4_0);
        case ORACLE_Q5_0: return (size_t) (n / 32) * sizeof(blk_q5_0);
        case ORACLE_Q8_0: return (size_t) (n / 32) * sizeof(blk_q8_0);
        case ORACLE_Q4_K: return (size_t) (n / 256) * sizeof(blk_q4_K);
        default: return 0;
    }
}

/* get_scale_min_k4 (ggml-quants.c:880-887) */
static void scale_min_k4(int j, const uint8_t * q, uint8_t * d, uint8_t * m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (uint8_t) ((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)); *m = (uint8_t) ((q[j + 4] >> 4) | ((q[j] >> 6) << 4)); }
}

void oracle_dequantize_row(int type, const void * blocks, float * y, int64_t n) {
    if (type == ORACLE_Q4_0) {                                            /* ggml-quants.c:459-477 */
        const blk_q4_0 * b = (const blk_q4_0 *) blocks;
        for (int64_t i = 0; i < n / 32; i++) {
            const float d = oracle_f16_to_f32(b[i].d);
            for (int j = 0; j < 16; j++) {
                y[i*32 + j]      = (float) ((b[i].qs[j] & 0x0F) - 8) * d;
                y[i*32 + j + 16] = (float) ((b[i].qs[j] >> 4)   - 8) * d;
            }
        }
    } else if (type == ORACLE_Q5_0) {                                     /* ggml-quants.c:500-524 */
        const blk_q5_0 * b = (const blk_q5_0 *) blocks;
        for (int64_t i = 0; i < n / 32; i++) {
            const float d = oracle_f16_to_f32(b[i].d);
            uint32_t qh; memcpy(&qh, b[i].qh, 4);
            for (int j = 0; j < 16; j++) {
                const uint8_t xh0 = (uint8_t) (((qh >> (j + 0)) << 4) & 0x10);
                const uint8_t xh1 = (uint8_t) ((qh >> (j + 12)) & 0x10);
                y[i*32 + j]      = (float) (((b[i].qs[j] & 0x0F) | xh0) - 16) * d;
                y[i*32 + j + 16] = (float) (((b[i].qs[j] >> 4)   | xh1) - 16) * d;
            }
        }
    } else if (type == ORACLE_Q8_0) {                                     /* ggml-quants.c:553-567 */
        const blk_q8_0 * b = (const blk_q8_0 *) blocks;
        for (int64_t i = 0; i < n / 32; i++) {
            const float d = oracle_f16_to_f32(b[i].d);
            for (int j = 0; j < 32; j++) y[i*32 + j] = (float) b[i].qs[j] * d;
        }
    } else if (type == ORACLE_Q4_K) {                                     /* ggml-quants.c:1529-1551 */
        const blk_q4_K * b = (const blk_q4_K *) blocks;
        for (int64_t i = 0; i < n / 256; i++) {
            const float d = oracle_f16_to_f32(b[i].d), mn = oracle_f16_to_f32(b[i].dmin);
            const uint8_t * q = b[i].qs;
            int is = 0;
            for (int j = 0; j < 256; j += 64) {
                uint8_t sc, m;
                scale_min_k4(is + 0, b[i].scales, &sc, &m); const float d1 = d * sc, m1 = mn * m;
                scale_min_k4(is + 1, b[i].scales, &sc, &m); const float d2 = d * sc, m2 = mn * m;
                for (int l = 0; l < 32; l++) y[i*256 + j + l]      = d1 * (float) (q[l] & 0xF) - m1;
                for (int l = 0; l < 32; l++) y[i*256 + j + 32 + l] = d2 * (float) (q[l] >> 4)  - m2;
                q += 32; is += 2;
            }
        }
    } else if (type == ORACLE_F16) {
        const uint16_t * h = (const uint16_t *) blocks;
        for (int64_t i = 0; i < n; i++) y[i] = oracle_f16_to_f32(h[i]);
    } else if (type == ORACLE_F32) {
        memcpy(y, blocks, (size_t) n * 4);
    }
}

#define OR_MIN(a, b) ((a) < (b) ? (a) : (b))


/* ---- Q4_K weight quantizer (quantize_row_q4_K_ref, ggml-quants.c:1457-1527, with make_qkx2_quants :799-878) ----------
 * Per 32 values: fit x ~ scale*q + min (q in 0..15, min <= 0) by weighted least squares, weights = rms(x) + |x|, trying the
 * 21 candidate grids iscale = (15 - 1 + 0.1*k)/(max - min); keep the one with the smallest weighted squared error.
 * Per 256 values: the 8 scales and 8 (negated) mins are themselves quantized to 6 bits against their maxima, the values are
 * re-rounded against the quantized scale/min.  Written for bit-equality with the reference built without FMA contraction
 * (upstream compiles ggml-base without -mfma; every product and sum below is a separate f32 operation in that order). */
static int q4k_rint(float v) { return (int) rintf(v); }                 /* nearest_int :621-626 (ties to even) */

static float q4k_fit32(const float * x, const float * wt, uint8_t * q, float * neg_min) {
    float lo = x[0], hi = x[0], sw = wt[0], swx = wt[0] * x[0];
    for (int i = 1; i < 32; i++) {
        if (x[i] < lo) lo = x[i];
        if (x[i] > hi) hi = x[i];
        sw += wt[i];
        swx += wt[i] * x[i];
    }
    if (lo > 0) lo = 0;
    if (hi == lo) { memset(q, 0, 32); *neg_min = -lo; return 0.0f; }
    float inv = 15 / (hi - lo), scale = 1 / inv, best = 0;
    for (int i = 0; i < 32; i++) {
        int l = q4k_rint(inv * (x[i] - lo));
        l = l < 0 ? 0 : (l > 15 ? 15 : l);
        q[i] = (uint8_t) l;
        float e = scale * q[i] + lo - x[i];
        e = e * e;
        best += wt[i] * e;
    }
    uint8_t trial[32];
    for (int step = 0; step <= 20; step++) {
        inv = (-1.f + 0.1f * step + 15) / (hi - lo);
        float sl = 0, sl2 = 0, sxl = 0;
        for (int i = 0; i < 32; i++) {
            int l = q4k_rint(inv * (x[i] - lo));
            l = l < 0 ? 0 : (l > 15 ? 15 : l);
            trial[i] = (uint8_t) l;
            const float w = wt[i];
            sl  += w * l;
            sl2 += w * l * l;
            sxl += w * l * x[i];
        }
        const float det = sw * sl2 - sl * sl;
        if (det > 0) {
            float sc = (sw * sxl - swx * sl) / det;
            float mn = (sl2 * swx - sl * sxl) / det;
            if (mn > 0) { mn = 0; sc = sxl / sl2; }
            float err = 0;
            for (int i = 0; i < 32; i++) {
                float e = sc * trial[i] + mn - x[i];
                e = e * e;
                err += wt[i] * e;
            }
            if (err < best) { memcpy(q, trial, 32); best = err; scale = sc; lo = mn; }
        }
    }
    *neg_min = -lo;
    return scale;
}

static void q4k_quantize_superblock(const float * x, blk_q4_K * y) {
    uint8_t q[256];
    float scales[8], mins[8], wt[32];
    float top_scale = 0, top_min = 0;
    for (int j = 0; j < 8; j++) {
        float ss = 0;
        for (int l = 0; l < 32; l++) ss += x[32*j + l] * x[32*j + l];
        const float rms = sqrtf(ss / 32);
        for (int l = 0; l < 32; l++) wt[l] = rms + fabsf(x[32*j + l]);
        scales[j] = q4k_fit32(x + 32*j, wt, q + 32*j, &mins[j]);
        if (scales[j] > top_scale) top_scale = scales[j];
        if (mins[j] > top_min) top_min = mins[j];
    }
    const float is = top_scale > 0 ? 63.f / top_scale : 0.f, im = top_min > 0 ? 63.f / top_min : 0.f;
    memset(y->scales, 0, 12);
    for (int j = 0; j < 8; j++) {                                       /* 6-bit packing, get_scale_min_k4's inverse (:880-887) */
        uint8_t ls = (uint8_t) q4k_rint(is * scales[j]), lm = (uint8_t) q4k_rint(im * mins[j]);
        if (ls > 63) ls = 63;
        if (lm > 63) lm = 63;
        if (j < 4) { y->scales[j] |= ls; y->scales[j + 4] |= lm; }
        else {
            y->scales[j + 4] |= (uint8_t) ((ls & 0xF) | ((lm & 0xF) << 4));
            y->scales[j - 4] |= (uint8_t) ((ls >> 4) << 6);
            y->scales[j]     |= (uint8_t) ((lm >> 4) << 6);
        }
    }
    y->d    = oracle_f32_to_f16(top_scale / 63.f);
    y->dmin = oracle_f32_to_f16(top_min / 63.f);
    for (int j = 0; j < 8; j++) {                                       /* re-round against the quantized scale / min */
        int sc, m;
        if (j < 4) { sc = y->scales[j] & 63; m = y->scales[j + 4] & 63; }
        else { sc = (y->scales[j + 4] & 0xF) | ((y->scales[j - 4] >> 6) << 4); m = (y->scales[j + 4] >> 4) | ((y->scales[j] >> 6) << 4); }
        const float d = oracle_f16_to_f32(y->d) * sc;
        if (!d) continue;
        const float dm = oracle_f16_to_f32(y->dmin) * m;
        for (int l = 0; l < 32; l++) {
            int v = q4k_rint((x[32*j + l] + dm) / d);
            q[32*j + l] = (uint8_t) (v < 0 ? 0 : (v > 15 ? 15 : v));
        }
    }
    for (int c = 0; c < 4; c++)
        for (int l = 0; l < 32; l++) y->qs[32*c + l] = (uint8_t) (q[64*c + l] | (q[64*c + 32 + l] << 4));
}

void oracle_quantize_row_ref(int type, const float * x, void * blocks, int64_t n) {
    if (type == ORACLE_Q4_0 || type == ORACLE_Q5_0) {                     /* ggml-quants.c:113-148, :187-230 */
        const int lv = type == ORACLE_Q4_0 ? 8 : 16;
        for (int64_t i = 0; i < n / 32; i++) {
            float amax = 0.0f, max = 0.0f;
            for (int j = 0; j < 32; j++) { const float v = x[i*32 + j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
            const float d = max / (float) -lv;
            const float id = d ? 1.0f / d : 0.0f;
            if (type == ORACLE_Q4_0) {
                blk_q4_0 * b = (blk_q4_0 *) blocks + i;
                b->d = oracle_f32_to_f16(d);
                for (int j = 0; j < 16; j++) {
                    const float x0 = x[i*32 + j] * id, x1 = x[i*32 + 16 + j] * id;
                    const uint8_t xi0 = (uint8_t) OR_MIN(15, (int8_t) (x0 + 8.5f)), xi1 = (uint8_t) OR_MIN(15, (int8_t) (x1 + 8.5f));
                    b->qs[j] = (uint8_t) (xi0 | (xi1 << 4));
                }
            } else {
                blk_q5_0 * b = (blk_q5_0 *) blocks + i;
                b->d = oracle_f32_to_f16(d);
                uint32_t qh = 0;
                for (int j = 0; j < 16; j++) {
                    const float x0 = x[i*32 + j] * id, x1 = x[i*32 + 16 + j] * id;
                    const uint8_t xi0 = (uint8_t) OR_MIN(31, (int8_t) (x0 + 16.5f)), xi1 = (uint8_t) OR_MIN(31, (int8_t) (x1 + 16.5f));
                    b->qs[j] = (uint8_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
                    qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
                    qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
                }
                memcpy(b->qh, &qh, 4);
            }
        }
    } else if (type == ORACLE_Q8_0) {                                     /* ggml-quants.c:276-299 (roundf: ties away) */
        for (int64_t i = 0; i < n / 32; i++) {
            blk_q8_0 * b = (blk_q8_0 *) blocks + i;
            float amax = 0.0f;
            for (int j = 0; j < 32; j++) amax = fmaxf(amax, fabsf(x[i*32 + j]));
            const float d = amax / 127.0f, id = d ? 1.0f / d : 0.0f;
            b->d = oracle_f32_to_f16(d);
            for (int j = 0; j < 32; j++) b->qs[j] = (int8_t) roundf(x[i*32 + j] * id);
        }
    } else if (type == ORACLE_Q4_K) {
        for (int64_t i = 0; i < n / 256; i++) q4k_quantize_superblock(x + i*256, (blk_q4_K *) blocks + i);
    } else if (type == ORACLE_F16) {
        uint16_t * h = (uint16_t *) blocks;
        for (int64_t i = 0; i < n; i++) h[i] = oracle_f32_to_f16(x[i]);
    }
}

/* AVX2 activation quantizer (ggml-cpu/arch/x86/quants.c:302-398): d = amax/127, multiplier 127/amax,
 * _mm256_round_ps(NEAREST) = ties-to-even, saturating packs (no saturation can occur: |x*id| <= 127) */
void oracle_quantize_row_q8_0(const float * x, void * blocks, int64_t n) {
    for (int64_t i = 0; i < n / 32; i++) {
        blk_q8_0 * b = (blk_q8_0 *) blocks + i;
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) amax = fmaxf(amax, fabsf(x[i*32 + j]));
        const float d = amax / 127.0f;
        const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
        b->d = oracle_f32_to_f16(d);
        for (int j = 0; j < 32; j++) b->qs[j] = (int8_t) (int) rintf(x[i*32 + j] * id);
    }
}

/* nearest_int (ggml-quants.c:621-626): round-to-nearest-even through the 1.5*2^23 trick == rintf for |v| < 2^22 */
void oracle_quantize_row_q8_K(const float * x, void * blocks, int64_t n) {      /* ggml-quants.c:2768-2805 */
    for (int64_t i = 0; i < n / 256; i++) {
        blk_q8_K * b = (blk_q8_K *) blocks + i;
        const float * xb = x + i*256;
        float max = 0, amax = 0;
        for (int j = 0; j < 256; j++) { const float ax = fabsf(xb[j]); if (ax > amax) { amax = ax; max = xb[j]; } }
        if (!amax) { b->d = 0; memset(b->qs, 0, 256); memset(b->bsums, 0, sizeof(b->bsums)); continue; }
        const float iscale = -127.f / max;
        for (int j = 0; j < 256; j++) { const int v = (int) rintf(iscale * xb[j]); b->qs[j] = (int8_t) OR_MIN(127, v); }
        for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 16; k++) s += b->qs[j*16 + k]; b->bsums[j] = (int16_t) s; }
        b->d = 1 / iscale;
    }
}

float oracle_vec_dot(int type, int64_t n, const void * w, const void * act) {
    float sumf = 0.0f;
    if (type == ORACLE_Q4_0 || type == ORACLE_Q5_0 || type == ORACLE_Q8_0) {
        const blk_q8_0 * y = (const blk_q8_0 *) act;
        for (int64_t i = 0; i < n / 32; i++) {
            int sumi = 0; float dw;
            if (type == ORACLE_Q4_0) {                                   /* quants.c:225-259 */
                const blk_q4_0 * b = (const blk_q4_0 *) w + i; dw = oracle_f16_to_f32(b->d);
                for (int j = 0; j < 16; j++) sumi += ((b->qs[j] & 0x0F) - 8) * y[i].qs[j] + ((b->qs[j] >> 4) - 8) * y[i].qs[j + 16];
            } else if (type == ORACLE_Q5_0) {                            /* quants.c:365-406 */
                const blk_q5_0 * b = (const blk_q5_0 *) w + i; dw = oracle_f16_to_f32(b->d);
                uint32_t qh; memcpy(&qh, b->qh, 4);
                for (int j = 0; j < 16; j++) {
                    const int x0 = (int) (((b->qs[j] & 0x0F) | (((qh >> j) & 1) << 4))) - 16;
                    const int x1 = (int) (((b->qs[j] >> 4)   | (((qh >> (j + 16)) & 1) << 4))) - 16;
                    sumi += x0 * y[i].qs[j] + x1 * y[i].qs[j + 16];
                }
            } else {                                                     /* quants.c:451-479 */
                const blk_q8_0 * b = (const blk_q8_0 *) w + i; dw = oracle_f16_to_f32(b->d);
                for (int j = 0; j < 32; j++) sumi += b->qs[j] * y[i].qs[j];
            }
            /* AVX2: acc = fmadd(broadcast(d_w*d_x), cvt(int sums), acc) — arch/x86/quants.c:1142-1181 */
            sumf = fmaf(dw * oracle_f16_to_f32(y[i].d), (float) sumi, sumf);
        }
    } else if (type == ORACLE_Q4_K) {                                    /* quants.c:696-769, arch/x86/quants.c:2038-2110 */
        const blk_q4_K * x = (const blk_q4_K *) w; const blk_q8_K * y = (const blk_q8_K *) act;
        float summ = 0.0f;
        for (int64_t i = 0; i < n / 256; i++) {
            const float d = y[i].d * oracle_f16_to_f32(x[i].d), dmin = -y[i].d * oracle_f16_to_f32(x[i].dmin);
            int sumi = 0, summi = 0;
            for (int j = 0; j < 8; j++) {
                uint8_t sc, m; scale_min_k4(j, x[i].scales, &sc, &m);
                const uint8_t * q4 = x[i].qs + (j >> 1) * 32; const int8_t * q8 = y[i].qs + j * 32;
                int s = 0;
                for (int l = 0; l < 32; l++) s += (int) ((j & 1) ? (q4[l] >> 4) : (q4[l] & 0xF)) * q8[l];
                sumi += sc * s;
                summi += m * (y[i].bsums[2*j] + y[i].bsums[2*j + 1]);
            }
            sumf = fmaf(d, (float) sumi, sumf);
            summ = fmaf(dmin, (float) summi, summ);
        }
        sumf += summ;
    } else if (type == ORACLE_F16) {                                     /* ggml_vec_dot_f16: f32 fma of converted halves */
        const uint16_t * a = (const uint16_t *) w, * b = (const uint16_t *) act;
        for (int64_t i = 0; i < n; i++) sumf = fmaf(oracle_f16_to_f32(a[i]), oracle_f16_to_f32(b[i]), sumf);
    } else if (type == ORACLE_F32) {
        const float * a = (const float *) w, * b = (const float *) act;
        for (int64_t i = 0; i < n; i++) sumf = fmaf(a[i], b[i], sumf);
    }
    return sumf;
}

void oracle_mul_mat(int type, const void * w, const float * x, float * dst, int64_t K, int64_t N, int64_t T) {
    const size_t wrow = oracle_row_size(type, K);
    size_t arow; void * act;
    if (type == ORACLE_Q4_K) arow = (size_t) (K / 256) * sizeof(blk_q8_K);
    else if (type == ORACLE_F16) arow = (size_t) K * 2;
    else if (type == ORACLE_F32) arow = (size_t) K * 4;
    else arow = (size_t) (K / 32) * sizeof(blk_q8_0);
    act = malloc(arow);
    for (int64_t t = 0; t < T; t++) {
        const float * xr = x + t*K;                                       /* src1 -> vec_dot_type, ggml-cpu.c:1322-1357 */
        if (type == ORACLE_Q4_K) oracle_quantize_row_q8_K(xr, act, K);
        else if (type == ORACLE_F16) { uint16_t * h = (uint16_t *) act; for (int64_t k = 0; k < K; k++) h[k] = oracle_f32_to_f16(xr[k]); }
        else if (type == ORACLE_F32) memcpy(act, xr, arow);
        else oracle_quantize_row_q8_0(xr, act, K);
        for (int64_t n = 0; n < N; n++) dst[t*N + n] = oracle_vec_dot(type, K, (const char *) w + (size_t) n*wrow, act);
    }
    free(act);
}

void oracle_norm(const float * x, float * y, int64_t n, int64_t nrows, float eps) {      /* ops.cpp:3698-3765 */
    for (int64_t r = 0; r < nrows; r++) {
        const float * xr = x + r*n; float * yr = y + r*n;
        double sum = 0.0;                                                 /* ggml_vec_sum_f32 accumulates in ggml_float */
        for (int64_t i = 0; i < n; i++) sum += (double) xr[i];
        const float mean = (float) sum / n;
        double var = 0.0;
        for (int64_t i = 0; i < n; i++) { const float v = xr[i] - mean; yr[i] = v; var += (double) (v * v); }
        const float variance = (float) (var / n);
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int64_t i = 0; i < n; i++) yr[i] *= scale;
    }
}

static uint16_t g_gelu_tab[65536]; static int g_gelu_init = 0;
void oracle_gelu(const float * x, float * y, int64_t n) {                                  /* vec.h:968-1000 */
    if (!g_gelu_init) {
        for (int i = 0; i < 65536; i++) {
            const float f = oracle_f16_to_f32((uint16_t) i);
            /* the reference is built with gcc's default -ffp-contract=fast: (A*x)*x + 1 becomes one fma.  With the
             * explicit fmaf all 65536 entries equal ggml_table_gelu_f16 of the AVX2 reference build (tests/golden). */
            const float inner = fmaf(0.044715f*f, f, 1.0f);
            g_gelu_tab[i] = oracle_f32_to_f16(0.5f*f*(1.0f + tanhf(0.79788456080286535587989211986876f*f*inner)));
        }
        g_gelu_init = 1;
    }
    for (int64_t i = 0; i < n; i++) {
        if (x[i] <= -10.0f) y[i] = 0.0f;
        else if (x[i] >= 10.0f) y[i] = x[i];
        else y[i] = oracle_f16_to_f32(g_gelu_tab[oracle_f32_to_f16(x[i])]);
    }
}

void oracle_soft_max(const float * x, const float * mask, float * y, int64_t n, int64_t nrows, float scale) {   /* ops.cpp:5455-5565 */
    for (int64_t r = 0; r < nrows; r++) {
        const float * xr = x + r*n; float * yr = y + r*n;
        float mx = -INFINITY;
        for (int64_t i = 0; i < n; i++) { yr[i] = xr[i]*scale + (mask ? mask[r*n + i] : 0.0f); if (yr[i] > mx) mx = yr[i]; }
        double sum = 0.0;
        for (int64_t i = 0; i < n; i++) { const float e = expf(yr[i] - mx); yr[i] = e; sum += (double) e; }
        const float inv = (float) (1.0 / sum);
        for (int64_t i = 0; i < n; i++) yr[i] *= inv;
    }
}

void oracle_im2col_1d_f16(const float * x, uint16_t * dst, int64_t IW, int64_t IC, int64_t OW, int KW, int s0, int p0, int d0) {
    for (int64_t ow = 0; ow < OW; ow++)                                                    /* ops.cpp:6486-6505 */
        for (int64_t ic = 0; ic < IC; ic++)
            for (int k = 0; k < KW; k++) {
                const int64_t iw = ow*s0 + (int64_t) k*d0 - p0;
                dst[ow*(IC*KW) + ic*KW + k] = (iw < 0 || iw >= IW) ? 0 : oracle_f32_to_f16(x[ic*IW + iw]);
            }
}

void oracle_rope(const float * x, const int32_t * pos, float * y, int64_t ne0, int64_t n_head, int64_t n_pos, int n_dims, int mode,
                 int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    /* ggml_rope_yarn_corr_dims (ggml.c:4371-4383) */
    const float c0 = n_dims * logf(n_ctx_orig / (beta_fast * 2 * (float) M_PI)) / (2 * logf(freq_base));
    const float c1 = n_dims * logf(n_ctx_orig / (beta_slow * 2 * (float) M_PI)) / (2 * logf(freq_base));
    const float lo = fmaxf(0, floorf(c0)), hi = fminf((float) (n_dims - 1), ceilf(c1));
    float * cache = (float *) malloc((size_t) ne0 * sizeof(float));
    for (int64_t p = 0; p < n_pos; p++) {
        float theta = (float) pos[p];                                                     /* ggml_rope_cache_init, ops.cpp:5845-5860 */
        for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
            const float theta_extrap = theta;
            float theta_interp = freq_scale * theta_extrap, th = theta_interp, mscale = attn_factor;
            if (ext_factor != 0.0f) {
                const float yv = ((float) (i0 / 2) - lo) / fmaxf(0.001f, hi - lo);
                const float ramp = (1 - fminf(1, fmaxf(0, yv))) * ext_factor;
                th = theta_interp * (1 - ramp) + theta_extrap * ramp;
                mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
            }
            cache[i0] = cosf(th) * mscale; cache[i0 + 1] = sinf(th) * mscale;
            theta *= theta_scale;
        }
        for (int64_t h = 0; h < n_head; h++) {
            const float * xr = x + (p*n_head + h)*ne0; float * yr = y + (p*n_head + h)*ne0;
            for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
                if (i0 < n_dims) {
                    const int64_t ia = mode == 0 ? i0 : i0/2, ib = mode == 0 ? i0 + 1 : i0/2 + n_dims/2;
                    const float x0 = xr[ia], x1 = xr[ib];
                    yr[ia] = x0*cache[i0] - x1*cache[i0 + 1];
                    yr[ib] = x0*cache[i0 + 1] + x1*cache[i0];
                } else { yr[i0] = xr[i0]; yr[i0 + 1] = xr[i0 + 1]; }
            }
        }
    }
    free(cache);
}

void oracle_flash_attn(const float * q, const uint16_t * k, const uint16_t * v, const uint16_t * mask, float * dst,
                       int64_t D, int64_t T, int64_t H, int64_t n_kv, float scale) {       /* ops.cpp:8479-8715 */
    uint16_t * qh = (uint16_t *) malloc((size_t) D * 2);
    uint16_t * acc = (uint16_t *) malloc((size_t) D * 2);
    for (int64_t t = 0; t < T; t++) for (int64_t h = 0; h < H; h++) {
        const float * qp = q + (t*H + h)*D;
        for (int64_t d = 0; d < D; d++) { qh[d] = oracle_f32_to_f16(qp[d]); acc[d] = 0; }
        float S = 0.0f, M = -INFINITY;
        for (int64_t ic = 0; ic < n_kv; ic++) {
            const float mv = mask ? oracle_f16_to_f32(mask[t*n_kv + ic]) : 0.0f;
            if (mv == -INFINITY) continue;
            float s = 0.0f;
            const uint16_t * kp = k + (ic*H + h)*D, * vp = v + (ic*H + h)*D;
            for (int64_t d = 0; d < D; d++) s = fmaf(oracle_f16_to_f32(kp[d]), oracle_f16_to_f32(qh[d]), s);
            s = s*scale + mv;
            const float Mold = M; float ms = 1.0f, vs = 1.0f;
            if (s > M) { M = s; ms = expf(Mold - M); for (int64_t d = 0; d < D; d++) acc[d] = oracle_f32_to_f16(oracle_f16_to_f32(acc[d]) * ms); }
            else vs = expf(s - M);
            for (int64_t d = 0; d < D; d++) acc[d] = oracle_f32_to_f16(oracle_f16_to_f32(acc[d]) + oracle_f16_to_f32(vp[d]) * vs);   /* ggml_vec_mad_f16 */
            S = S*ms + vs;
        }
        const float inv = S == 0.0f ? 0.0f : 1.0f / S;
        for (int64_t d = 0; d < D; d++) dst[(t*H + h)*D + d] = oracle_f16_to_f32(acc[d]) * inv;
    }
    free(qh); free(acc);
}

/* ------------------------------------------------------------------------------------------------------------------
 * ggml_flash_attn_ext as the AVX2 + FMA + F16C CPU build DISPATCHES it (ggml-cpu/ops.cpp:9077-9230).  The arithmetic
 * depends on the shape, three paths:
 *   T == 1 and n_kv >= 512 : split-KV — the key range is cut into nth chunks (one per thread), each runs the vec path
 *                            below over its keys, the f32 partials (M, S, VKQ) are merged by
 *                            ggml_flash_attn_ext_reduce_partials (ops.cpp:8992-9075).  RESULT DEPENDS ON nth.
 *   T >= 64                : tiled path (ops.cpp:8717-8990): q stays F32, K/V tiles converted to F32, scores and P.V by
 *                            simd_gemm (sequential fma over the inner index, simd-gemm.h:24-58), softmax of a 64-key tile
 *                            with ggml_v_expf (vec.h:1215-1252) and the 8-lane sum order of ggml_vec_soft_max_f32
 *                            (vec.cpp:541-551); F32 accumulation.
 *   otherwise              : vec path `one_chunk` (ops.cpp:8479-8715): q -> F16, score by ggml_vec_dot_f16 (vec.cpp:264,
 *                            GGML_F16_STEP 32 / EPR 8, reduction simd-mappings.h:602-620), sequential online softmax with
 *                            libm expf, V accumulated in F16 (ggml_vec_mad_f16 / ggml_vec_scale_f16: f32 fma, store as f16).
 * Restated lane by lane in the SIMD order so that the result is bit-identical to the reference build in oracle/_ref
 * (pinned by tests/golden/ops.npz: golden_flash_attn, golden_fattn_split, golden_fattn_tiled).
 * ------------------------------------------------------------------------------------------------------------------ */
static float fa_bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t fa_f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* one lane of ggml_v_expf (AVX2 variant), vec.h:1215-1252 */
float oracle_v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = fa_f_bits(z) << 23;
    const float k = fa_bits_f(e + fa_f_bits(1.0f));
    const int c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, 0x1.ffffecp-1f * b);
    if (!c) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = fa_bits_f(g + 0x7f000000u);
    const float s2 = fa_bits_f(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}

/* ggml_vec_dot_f16 for n % 32 == 0: four 8-lane f32 accumulators, fma per 32-element step, then x0+=x2, x1+=x3, x0+=x1,
 * low half + high half, two horizontal adds */
static float fa_dot_f16(const uint16_t * x, const uint16_t * y, int64_t n) {
    float s[4][8];
    for (int j = 0; j < 4; j++) for (int l = 0; l < 8; l++) s[j][l] = 0.0f;
    for (int64_t i = 0; i < n; i += 32)
        for (int j = 0; j < 4; j++) for (int l = 0; l < 8; l++)
            s[j][l] = fmaf(oracle_f16_to_f32(x[i + j*8 + l]), oracle_f16_to_f32(y[i + j*8 + l]), s[j][l]);
    float v[8], t0[4];
    for (int l = 0; l < 8; l++) { const float a = s[0][l] + s[2][l], b = s[1][l] + s[3][l]; v[l] = a + b; }
    for (int i = 0; i < 4; i++) t0[i] = v[i] + v[i + 4];
    return (t0[0] + t0[1]) + (t0[2] + t0[3]);
}

/* vec path over keys [ic0, ic1) of one (query, head): leaves M, S and the F32 image of the F16 accumulator */
static void fa_one_chunk(const uint16_t * qh, const uint16_t * k, const uint16_t * v, const uint16_t * mrow, int64_t D, int64_t H, int64_t h,
                         int64_t ic0, int64_t ic1, float scale, float * M_out, float * S_out, float * vkq32) {
    uint16_t acc[256];
    for (int64_t d = 0; d < D; d++) acc[d] = 0;
    float S = 0.0f, M = -INFINITY;
    for (int64_t ic = ic0; ic < ic1; ic++) {
        const float mv = mrow ? oracle_f16_to_f32(mrow[ic]) : 0.0f;
        if (mv == -INFINITY) continue;
        const uint16_t * kp = k + (ic*H + h)*D, * vp = v + (ic*H + h)*D;
        float s = fa_dot_f16(kp, qh, D);
        s = s*scale;
        s += mv;
        const float Mold = M;
        float ms = 1.0f, vs = 1.0f;
        if (s > M) {
            M = s; ms = expf(Mold - M);
            for (int64_t d = 0; d < D; d++) acc[d] = oracle_f32_to_f16(oracle_f16_to_f32(acc[d]) * ms);          /* ggml_vec_scale_f16 */
        } else vs = expf(s - M);
        for (int64_t d = 0; d < D; d++) acc[d] = oracle_f32_to_f16(fmaf(oracle_f16_to_f32(vp[d]), vs, oracle_f16_to_f32(acc[d])));   /* ggml_vec_mad_f16 */
        S = S*ms; S = S + vs;                      /* two roundings: gcc specialises the statement per branch (ms == 1 or vs == 1), no fma in the reference build */
    }
    for (int64_t d = 0; d < D; d++) vkq32[d] = oracle_f16_to_f32(acc[d]);
    *M_out = M; *S_out = S;
}

/* tiled path for ONE query row (rows of a tile are independent): 64-key tiles */
static void fa_tiled_row(const float * q, const uint16_t * k, const uint16_t * v, const uint16_t * mrow, int64_t D, int64_t H, int64_t h,
                         int64_t n_kv, float scale, float * out) {
    float vkq[256], kq[64], vstale[64][256];
    for (int64_t d = 0; d < D; d++) vkq[d] = 0.0f;
    memset(vstale, 0, sizeof(vstale));
    float S = 0.0f, M = -INFINITY;
    for (int64_t ic = 0; ic < n_kv; ic += 64) {
        const int kvt = (int) (n_kv - ic < 64 ? n_kv - ic : 64);
        if (mrow) {
            int can_skip = 1;
            for (int tk = 0; tk < kvt; tk++) if (oracle_f16_to_f32(mrow[ic + tk]) != -INFINITY) can_skip = 0;
            /* (the reference skips a tile only when EVERY row of the 64-query tile is masked; for one row the arithmetic below
             * gives the same result: tile_max == -inf => no contribution) */
            (void) can_skip;
        }
        for (int tk = 0; tk < 64; tk++) {
            if (tk >= kvt) { kq[tk] = -INFINITY; continue; }
            const uint16_t * kp = k + ((ic + tk)*H + h)*D;
            float a = 0.0f;
            for (int64_t d = 0; d < D; d++) a = fmaf(oracle_f16_to_f32(kp[d]), q[d], a);                 /* simd_gemm: sequential fma over dk */
            a = a * scale;
            if (mrow) a = a + oracle_f16_to_f32(mrow[ic + tk]);
            kq[tk] = a;
        }
        for (int tk = 0; tk < kvt; tk++) { const uint16_t * vp = v + ((ic + tk)*H + h)*D; for (int64_t d = 0; d < D; d++) vstale[tk][d] = oracle_f16_to_f32(vp[d]); }
        float tmax = -INFINITY;
        for (int tk = 0; tk < 64; tk++) if (kq[tk] > tmax) tmax = kq[tk];
        if (tmax == -INFINITY) continue;
        const float Mold = M, Mnew = fmaxf(Mold, tmax);
        if (Mnew > Mold) { const float ms = expf(Mold - Mnew); for (int64_t d = 0; d < D; d++) vkq[d] *= ms; S *= ms; }
        M = Mnew;
        double sum = 0.0;
        for (int g = 0; g < 8; g++) {                                  /* ggml_vec_soft_max_f32: 8 lanes at a time */
            float p[8];
            for (int l = 0; l < 8; l++) { p[l] = oracle_v_expf(kq[g*8 + l] - Mnew); kq[g*8 + l] = p[l]; }
            const float a0 = p[0] + p[4], a1 = p[1] + p[5], a2 = p[2] + p[6], a3 = p[3] + p[7];
            sum += (double) ((a0 + a2) + (a1 + a3));
        }
        S = (float) ((double) S + sum);
        for (int tk = 0; tk < 64; tk++) for (int64_t d = 0; d < D; d++) vkq[d] = fmaf(vstale[tk][d], kq[tk], vkq[d]);    /* simd_gemm: sequential fma over keys */
    }
    const float inv = S == 0.0f ? 0.0f : 1.0f / S;
    for (int64_t d = 0; d < D; d++) out[d] = vkq[d] * inv;
}

void oracle_flash_attn_ext(const float * q, const uint16_t * k, const uint16_t * v, const uint16_t * mask, float * dst,
                           int64_t D, int64_t T, int64_t H, int64_t n_kv, float scale, int nth) {
    if (D > 256 || D % 32) return;
    uint16_t qh[256];
    float part[256];
    if (nth < 1) nth = 1;
    for (int64_t t = 0; t < T; t++) for (int64_t h = 0; h < H; h++) {
        const float * qp = q + (t*H + h)*D;
        float * out = dst + (t*H + h)*D;
        const uint16_t * mrow = mask ? mask + t*n_kv : NULL;
        if (T >= 64) { fa_tiled_row(qp, k, v, mrow, D, H, h, n_kv, scale, out); continue; }
        for (int64_t d = 0; d < D; d++) qh[d] = oracle_f32_to_f16(qp[d]);
        if (T == 1 && n_kv >= 512) {
            const int64_t chunk = (n_kv + nth - 1) / nth;
            float Mf = -INFINITY, Sf = 0.0f, fin[256];
            for (int64_t d = 0; d < D; d++) fin[d] = 0.0f;
            for (int c = 0; c < nth; c++) {
                const int64_t ic0 = (int64_t) c * chunk, ic1 = ic0 + chunk < n_kv ? ic0 + chunk : n_kv;
                if (ic0 >= n_kv) continue;
                float Mc, Sc;
                fa_one_chunk(qh, k, v, mrow, D, H, h, ic0, ic1, scale, &Mc, &Sc, part);
                if (Sc == 0.0f) continue;
                const float Mn = fmaxf(Mf, Mc);
                const float so = expf(Mf - Mn), sn = expf(Mc - Mn);
                /* a*b + c*d of the reference build (gcc -O3, -ffp-contract=fast): the second product is rounded, the first fused */
                for (int64_t d = 0; d < D; d++) fin[d] = fmaf(fin[d], so, part[d] * sn);
                Sf = fmaf(Sf, so, Sc * sn);
                Mf = Mn;
            }
            if (Sf != 0.0f) { const float inv = 1.0f / Sf; for (int64_t d = 0; d < D; d++) fin[d] *= inv; }
            for (int64_t d = 0; d < D; d++) out[d] = fin[d];
        } else {
            float M, S;
            fa_one_chunk(qh, k, v, mrow, D, H, h, 0, n_kv, scale, &M, &S, part);
            const float inv = S == 0.0f ? 0.0f : 1.0f / S;
            for (int64_t d = 0; d < D; d++) out[d] = part[d] * inv;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * log-mel front end: whisper's log_mel_spectrogram (src/whisper.cpp:3046-3283) — reflective 200-sample pad in front, 30 s of
 * zeros behind, Hann window 400 (periodic, :3031-3039), hop 160, the recursive radix-2 FFT that ends in 25-point DFTs
 * (:3046-3111, sin / cos table of 400 entries made with sinf / cosf :3022-3028), power spectrum, filterbank with f64 sums in
 * groups of four (:3141-3158), log10, clamp to (max - 8), (x + 4) / 4.  out[j * n_len + i]; n_len = (n + 480000) / 160.
 * ------------------------------------------------------------------------------------------------------------------ */
static float mel_sin[400], mel_cos[400], mel_hann[400];
static int mel_tab_ready = 0;
static void mel_tables(void) {
    if (mel_tab_ready) return;
    for (int i = 0; i < 400; i++) {
        const double theta = (2 * M_PI * i) / 400;
        mel_sin[i] = sinf((float) theta); mel_cos[i] = cosf((float) theta);
        mel_hann[i] = (float) (0.5 * (1.0 - cosf((float) ((2.0 * M_PI * i) / 400))));
    }
    mel_tab_ready = 1;
}
static void mel_dft(const float * in, int N, float * out) {
    const int step = 400 / N;
    for (int k = 0; k < N; k++) {
        float re = 0, im = 0;
        for (int n = 0; n < N; n++) { const int idx = (k * n * step) % 400; re += in[n]*mel_cos[idx]; im -= in[n]*mel_sin[idx]; }
        out[k*2 + 0] = re; out[k*2 + 1] = im;
    }
}
static void mel_fft(float * in, int N, float * out) {
    if (N == 1) { out[0] = in[0]; out[1] = 0; return; }
    const int h = N / 2;
    if (N - h*2 == 1) { mel_dft(in, N, out); return; }
    float * even = in + N;
    for (int i = 0; i < h; i++) even[i] = in[2*i];
    float * even_fft = out + 2*N;
    mel_fft(even, h, even_fft);
    float * odd = even;
    for (int i = 0; i < h; i++) odd[i] = in[2*i + 1];
    float * odd_fft = even_fft + N;
    mel_fft(odd, h, odd_fft);
    const int step = 400 / N;
    for (int k = 0; k < h; k++) {
        const int idx = k * step;
        const float re = mel_cos[idx], im = -mel_sin[idx];
        const float ro = odd_fft[2*k], io = odd_fft[2*k + 1];
        out[2*k]           = even_fft[2*k]     + re*ro - im*io;
        out[2*k + 1]       = even_fft[2*k + 1] + re*io + im*ro;
        out[2*(k + h)]     = even_fft[2*k]     - re*ro + im*io;
        out[2*(k + h) + 1] = even_fft[2*k + 1] - re*io - im*ro;
    }
}
int64_t oracle_log_mel_n_len(int64_t n_samples) { return (n_samples + 480000) / 160; }
void oracle_log_mel(const float * pcm, int64_t n, const float * filters, int64_t n_mel, float * out) {
    mel_tables();
    const int N = 400, step = 160, bins = 201;
    const int64_t total = n + 480000 + 400, n_len = (total - N) / step;
    float * pad = (float *) calloc((size_t) total, sizeof(float));
    memcpy(pad + 200, pcm, (size_t) n * sizeof(float));
    const int64_t nref = n - 1 < 200 ? (n - 1 < 0 ? 0 : n - 1) : 200;
    for (int64_t k = 1; k <= nref; k++) pad[200 - k] = pcm[k];                      /* reverse_copy(samples + 1, ...) */
    const int64_t ns = n + 200;                                                    /* the worker's n_samples */
    const int64_t live = ns / step + 1 < n_len ? ns / step + 1 : n_len;
    float * fin = (float *) calloc(2 * N, sizeof(float)), * fout = (float *) calloc(8 * N, sizeof(float));
    double * tmp = (double *) malloc((size_t) n_mel * n_len * sizeof(double));
    for (int64_t i = 0; i < n_len; i++) {
        if (i >= live) { for (int64_t j = 0; j < n_mel; j++) tmp[j*n_len + i] = log10(1e-10); continue; }
        const int64_t off = i * step;
        const int64_t m = N < ns - off ? N : ns - off;
        for (int64_t j = 0; j < m; j++) fin[j] = mel_hann[j] * pad[off + j];
        for (int64_t j = m < 0 ? 0 : m; j < 2*N; j++) fin[j] = 0.0f;
        mel_fft(fin, N, fout);
        for (int j = 0; j < bins; j++) fout[j] = fout[2*j]*fout[2*j] + fout[2*j + 1]*fout[2*j + 1];
        for (int64_t j = 0; j < n_mel; j++) {
            const float * f = filters + j * bins;
            double sum = 0.0; int k = 0;
            for (; k < bins - 3; k += 4) sum += fout[k]*f[k] + fout[k + 1]*f[k + 1] + fout[k + 2]*f[k + 2] + fout[k + 3]*f[k + 3];
            for (; k < bins; k++) sum += fout[k]*f[k];
            sum = log10(sum > 1e-10 ? sum : 1e-10);
            tmp[j*n_len + i] = (double) (float) sum;                              /* mel.data is float */
        }
    }
    double mmax = -1e20;
    for (int64_t i = 0; i < n_mel*n_len; i++) if (tmp[i] > mmax) mmax = tmp[i];
    mmax -= 8.0;
    for (int64_t i = 0; i < n_mel*n_len; i++) { double v = tmp[i] < mmax ? mmax : tmp[i]; out[i] = (float) ((v + 4.0) / 4.0); }
    free(pad); free(fin); free(fout); free(tmp);
}
