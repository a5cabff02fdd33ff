// Tile GEMM over the QUANTIZED operands: the reference's integer block dots on the CDNA4 matrix cores.
//
//   dst[m, t] = sum over 32-element blocks b of  dw[m,b] * dx[t,b] * ( sum_k qw[m,b,k] * qx[t,b,k] )
//
// The CPU path rounds src1 to the weight type's vec_dot_type and accumulates integer block dots with an f32 scale per block for EVERY
// column count (ggml-cpu/ggml-cpu.c:1322-1357; vec_dot q4_0 / q5_0 / q8_0 x q8_0: ggml-cpu/quants.c:225-259, :365-406, :451-479;
// q4_K x q8_K: :696-769).  The f16 MFMA path of gemm_mfma.hip approximates that with f16(d*q) products; here the integers themselves
// go through v_mfma_i32_32x32x32_i8 — K = 32 per instruction is exactly one quantization block — and every block's 32 x 32 tile of
// integer sums is folded into the f32 accumulators with acc = fma(float(sum), dw*dx, acc), the statement of the decoder mat-vecs
// (decode.hip).  What reaches dst are the CPU's own integer sums; only the f32 summation order over blocks differs (here: ascending
// block index; AVX2: eight partial lanes; oracle_mul_mat is matched to <= 1e-10 NMSE, tests/test_gpu_mmq.py).
//
//  * A = weights, read from HBM in the block-quantized planar layout (0.56 - 1.06 B/weight, mi355x_kernels.h) and unpacked PER
//    WORKGROUP to signed int8 on their way into an XOR-swizzled LDS tile — no f16 or int8 copy of any weight exists in HBM.
//  * B = activations already quantized to Q8_0 / Q8_K rows (qrows.h) by whatever produced them (LayerNorm, attention, the previous
//    GEMM's epilogue, or mi355x_prep_act); the tile copy is a straight 16-byte move.
//  * 256 threads = 4 waves in a 2 x 2 arrangement, wave tile (BMT/2) x (BNT/2), K-step = 128 elements (4 blocks), two LDS stages, ONE
//    barrier per K-step; the global loads of step k+1 are in flight while step k is computed.
//  * The fix-up is VALU work, and the VALU — 4 cycles per wave instruction, 128 for the naive 16 cvt + 16 fma against the 64 of the MFMA
//    pair (first GPU run: 57 us for fc1, the f16 path 46) — is what bounds the kernel.  So (i) the integer sums arrive AS FLOATS: the
//    MFMA's C operand is the constant 0x4B400000 in every element, i.e. D = bits(1.5 * 2^23) + sum, whose float value is 12582912 + sum
//    exactly for |sum| < 2^22 (Q8_0 x Q8_0: 32 * 127 * 127 < 2^19), and 12582912 is subtracted again — exactly, float(sum) bit for bit —
//    by a PACKED add; (ii) the fma is packed as well: 8 v_pk_add_f32 + 8 v_pk_fma_f32 = 64 cycles per MFMA pair.
//    The 32 x 32 tile of scale products dw[m]*dx[t] — 16 different
//    weight rows per lane — comes from the matrix cores as well: one v_mfma_f32_32x32x16_f16 whose A operand holds dw (f16, exact) in
//    K-slot 0 and whose B operand holds dx (an f16 value by construction of Q8_0) in K-slot 0, zeros elsewhere: a rank-1 product,
//    exact in f32.  Without it every lane would need 16 broadcast LDS reads and 16 multiplies per MFMA (SMF = false: the form Q4_K
//    uses, whose Q8_K activation scale is not an f16 value).
//  * Q4_K: unsigned nibbles in the A tile; per 32-element sub-block acc += float(sum) * (d*sc_j)[m] * d8[t] - (dmin*m_j)[m] * (d8*bsum_j)[t]
//    (the reference sums sc_j*sum_j in integers first; the difference is f32 rounding).
//  * XCD-aware tile order as in k_gemm_f16_ring (gemm_mfma.hip).
#include "common.h"
#include "qrows.h"
#include <atomic>

typedef int i32x4_t  __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

struct MmqArgs {
    const char * A; int64_t nbt;                                // planar quantized weights [K, M]; nbt = total blocks
    const int8_t * Bq; const float * Bd; const int * Bs;       // activation rows of x [K, T] (qrows.h)
    int M, K; int64_t T;
    char * dst; int64_t dst_nb1; int dst_f16;
    const float * bias; float scale; int has_scale; int gelu; int bias_t;
    const char * residual; int64_t res_nb1;
    const uint16_t * gelu_tab;
    int8_t * pq; float * pd; int prep_only;                    // epilogue also leaves the Q8_0 rows of the result (K' = M): the next GEMM's B
    int mt, nt, per, m_major;                                  // tile counts and XCD-aware tile order
};

#define MQ_KS 128                      // K elements per step = bytes per tile row
__device__ __forceinline__ int mq_off(int row, int slot) { return row*128 + ((slot ^ ((row >> 1) & 7)) << 4); }   // 16-byte slot of a [rows][128 B] tile

// ---- registers of one thread's share of the next A tile ----------------------------------------------------------------------------
// Q4_0 / Q5_0 / Q8_0: PA (row, block) pairs, pair j = (row (tid >> 2) + 64*j, block tid & 3 of the K-step): the four lanes of a row read
// its 4 blocks' quants, high bits and scales as ONE contiguous run each (64 / 16 / 8 bytes) — 16 runs per wave instruction instead of 32
// lines with one 16-byte piece used of each.
template <int WT, int PA> struct mq_aregs;
template <int PA> struct mq_aregs<MI355X_TYPE_Q4_0, PA> { uint4 q[PA]; uint16_t d[PA]; };
template <int PA> struct mq_aregs<MI355X_TYPE_Q5_0, PA> { uint4 q[PA]; uint32_t qh[PA]; uint16_t d[PA]; };
template <int PA> struct mq_aregs<MI355X_TYPE_Q8_0, PA> { uint4 q0[PA], q1[PA]; uint16_t d[PA]; };
// Q4_K: the thread's PA sub-blocks lie in ONE 64-element chunk of row tid / (4 / PA) (PA = 2: both nibbles of its 32 bytes; PA = 1: one of them)
template <int PA> struct mq_aregs<MI355X_TYPE_Q4_K, PA> { uint4 q0, q1; uint32_t dm; uint32_t sc[3]; };

// signed bytes from 5-bit values x = nib | bit << 4 (ggml-quants.c:500-524: w = x - 16): spread the four INVERTED high bits to byte
// masks, then OR 0xF0 into the bytes whose value is negative (x - 16 = 0xF0 | nib there, nib elsewhere)
__device__ __forceinline__ uint32_t q5_signed(uint32_t nib, uint32_t inv4) {
    const uint32_t t = ((inv4 & 0xFu) * 0x00204081u) & 0x01010101u;        // bit k -> bit 0 of byte k
    const uint32_t m = __builtin_amdgcn_perm(0u, 0u, t | 0x0C0C0C0Cu);     // v_perm_b32 selector 0x0C -> 0x00, 0x0D -> 0xFF: a byte mask without a multiply
    return nib | (m & 0xF0F0F0F0u);
}
// the same with the 4-bit -> byte-mask step read from a 16-entry table in LDS (lut[x] = 0xF0 in byte k where bit k of x is set): two
// VALU operations (extract, scale the index) and one ds_read_b32 instead of five — the unpack is VALU time the fix-up needs
__device__ __forceinline__ uint32_t q5_signed_lut(uint32_t nib, uint32_t inv, int shift, const uint32_t * lut) { return nib | lut[(inv >> shift) & 0xFu]; }
// signed bytes from nibbles (ggml-quants.c:459-477: w = nib - 8), four bytes at once: with bit 7 set as a guard the subtraction never
// borrows across bytes ((nib | 0x80) - 8 = 0x78 + nib), and flipping bit 7 back leaves nib - 8 in two's complement
// (nib >= 8: 0x80 + (nib - 8) -> nib - 8;  nib < 8: 0x78 + nib -> 0xF8 + nib)
__device__ __forceinline__ uint32_t q4_signed(uint32_t nib) { return ((nib | 0x80808080u) - 0x08080808u) ^ 0x80808080u; }

// unpack to int8 and store: tile rows of 128 bytes, block lb (0..3) of the K-step = slots 2*lb (elements 0..15), 2*lb + 1 (16..31);
// per-block f32 scales sA[lb][row] (Q4_K: d*sc_j, and mA[lb][row] = -(dmin*m_j))
// (Q4_0 / Q5_0 / Q8_0: pair i is tile row `row + 64*i`, block lb0; Q4_K: PA sub-blocks lb0 .. of tile row `row`)
template <int WT, int PA, int BMT, bool SMF>
__device__ __forceinline__ void mq_a_store(const mq_aregs<WT, PA> & r, char * At, float * sA, float * mA, int row, int lb0, int blk, const uint32_t * lut) {
    if constexpr (WT == MI355X_TYPE_Q4_K) {
        const float d = h2f((uint16_t) (r.dm & 0xFFFF)), dmin = h2f((uint16_t) (r.dm >> 16));
        const uint32_t w[8] = { r.q0.x, r.q0.y, r.q0.z, r.q0.w, r.q1.x, r.q1.y, r.q1.z, r.q1.w };
        #pragma unroll
        for (int i = 0; i < PA; i++) {
            const int j = (blk + i) & 7;                                   // sub-block of the super-block; nibble j & 1 of its chunk
            const int sh = (j & 1) * 4;
            uint32_t o[8];
            #pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (w[e] >> sh) & 0x0F0F0F0Fu;
            *(uint4 *) (At + mq_off(row, 2*(lb0 + i)))     = make_uint4(o[0], o[1], o[2], o[3]);
            *(uint4 *) (At + mq_off(row, 2*(lb0 + i) + 1)) = make_uint4(o[4], o[5], o[6], o[7]);
            int sc, m; q4k_scale_min_w(j, r.sc[0], r.sc[1], r.sc[2], sc, m);
            sA[(lb0 + i)*BMT + row] = d * (float) sc;                      // exact: 11-bit x 6-bit
            mA[(lb0 + i)*BMT + row] = -(dmin * (float) m);
        }
    } else {
        #pragma unroll
        for (int i = 0; i < PA; i++) {
            uint4 lo, hi;
            if constexpr (WT == MI355X_TYPE_Q8_0) { lo = r.q0[i]; hi = r.q1[i]; }
            else {
                const uint32_t w[4] = { r.q[i].x, r.q[i].y, r.q[i].z, r.q[i].w };
                uint32_t l[4], h[4];
                if constexpr (WT == MI355X_TYPE_Q5_0) {
                    const uint32_t inv = ~r.qh[i];
                    #pragma unroll
                    for (int e = 0; e < 4; e++) {
                        l[e] = q5_signed_lut(w[e] & 0x0F0F0F0Fu,        inv, 4*e,      lut);
                        h[e] = q5_signed_lut((w[e] >> 4) & 0x0F0F0F0Fu, inv, 16 + 4*e, lut);
                    }
                } else {
                    #pragma unroll
                    for (int e = 0; e < 4; e++) { l[e] = q4_signed(w[e] & 0x0F0F0F0Fu); h[e] = q4_signed((w[e] >> 4) & 0x0F0F0F0Fu); }
                }
                lo = make_uint4(l[0], l[1], l[2], l[3]); hi = make_uint4(h[0], h[1], h[2], h[3]);
            }
            const int ri = row + 64*i;
            *(uint4 *) (At + mq_off(ri, 2*lb0))     = lo;
            *(uint4 *) (At + mq_off(ri, 2*lb0 + 1)) = hi;
            if constexpr (SMF) ((uint32_t *) sA)[lb0*BMT + ri] = (uint32_t) r.d[i];      // the f16 itself: K-slot 0 of the rank-1 scale MFMA
            else sA[lb0*BMT + ri] = h2f(r.d[i]);
        }
    }
}

// ---- epilogue: bias / scale / GELU / residual, F32 or F16 store, optional Q8_0 rows of the result ---------------------------------
// C layout of the 32 x 32 MFMAs (any dtype): lane holds column t = lane & 31, rows (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
template <int MT, int NT>
__device__ __forceinline__ void mq_epilogue(const MmqArgs & a, floatx16 (&acc)[MT][NT], int mw /* first row of the wave */, int64_t nw /* first column */, int lane) {
    const bool vec_ok = (a.M % 4 == 0) && ((uintptr_t) a.dst % 16 == 0) && (a.dst_nb1 % 16 == 0) &&
                        (!a.residual || (((uintptr_t) a.residual % 16 == 0) && (a.res_nb1 % 16 == 0))) && (!a.bias || a.bias_t || ((uintptr_t) a.bias % 16 == 0));
    const int hf = lane >> 5;
    #pragma unroll
    for (int i = 0; i < MT; i++) {
        #pragma unroll
        for (int j = 0; j < NT; j++) {
            const int64_t t = nw + j*32 + (lane & 31);
            const int mb = mw + i*32;
            if (t >= a.T || mb >= a.M) continue;                        // the same for both lanes of a (lane, lane ^ 32) pair
            float v[16];
            float amax = 0.0f;
            #pragma unroll
            for (int g = 0; g < 4; g++) {
                const int m = mb + 8*g + 4*hf;
                float x[4] = { acc[i][j][4*g], acc[i][j][4*g+1], acc[i][j][4*g+2], acc[i][j][4*g+3] };
                if (vec_ok && m < a.M) {                                // m % 4 == 0 and M % 4 == 0  =>  m + 3 < M
                    if (a.bias) {
                        if (a.bias_t) { const float b = a.bias[t]; x[0] += b; x[1] += b; x[2] += b; x[3] += b; }
                        else { const float4 b = *(const float4 *) (a.bias + m); x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w; }
                    }
                    if (a.has_scale) { x[0] *= a.scale; x[1] *= a.scale; x[2] *= a.scale; x[3] *= a.scale; }
                    if (a.gelu) { x[0] = gelu_lut(x[0], a.gelu_tab); x[1] = gelu_lut(x[1], a.gelu_tab); x[2] = gelu_lut(x[2], a.gelu_tab); x[3] = gelu_lut(x[3], a.gelu_tab); }
                    if (a.residual) { const float4 r4 = *(const float4 *) (a.residual + t*a.res_nb1 + (int64_t) m*4); x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w; }
                    if (!a.prep_only) {
                        if (a.dst_f16) *(uint2 *) (a.dst + t*a.dst_nb1 + (int64_t) m*2) = make_uint2((uint32_t) f2h(x[0]) | ((uint32_t) f2h(x[1]) << 16), (uint32_t) f2h(x[2]) | ((uint32_t) f2h(x[3]) << 16));
                        else           *(float4 *) (a.dst + t*a.dst_nb1 + (int64_t) m*4) = make_float4(x[0], x[1], x[2], x[3]);
                    }
                } else {
                    #pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (m + e >= a.M) { x[e] = 0.0f; continue; }
                        if (a.bias) x[e] += a.bias[a.bias_t ? t : (int64_t) (m + e)];
                        if (a.has_scale) x[e] *= a.scale;
                        if (a.gelu) x[e] = gelu_lut(x[e], a.gelu_tab);
                        if (a.residual) x[e] += *(const float *) (a.residual + t*a.res_nb1 + (int64_t) (m + e)*4);
                        if (!a.prep_only) {
                            if (a.dst_f16) *(uint16_t *) (a.dst + t*a.dst_nb1 + (int64_t) (m + e)*2) = f2h(x[e]);
                            else           *(float *)    (a.dst + t*a.dst_nb1 + (int64_t) (m + e)*4) = x[e];
                        }
                    }
                }
                #pragma unroll
                for (int e = 0; e < 4; e++) { v[4*g + e] = x[e]; amax = fmaxf(amax, fabsf(x[e])); }
            }
            if (a.pq) {
                // The result is the activation matrix of the NEXT quantized GEMM (fc1 + GELU -> fc2, src/whisper.cpp:2224-2238): leave its
                // Q8_0 rows (quantize_row_q8_0, arch/x86/quants.c:302-398) straight from the accumulators.  A block = 32 consecutive
                // features of one token = the 16 registers of this lane and of lane ^ 32.  (host: M % 32 == 0)
                amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                const float d  = amax / 127.0f;
                const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int m = mb + 8*g + 4*hf;
                    const int q0 = (int) rintf(v[4*g]*id), q1 = (int) rintf(v[4*g+1]*id), q2 = (int) rintf(v[4*g+2]*id), q3 = (int) rintf(v[4*g+3]*id);
                    *(uint32_t *) (a.pq + t*a.M + m) = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
                }
                if (hf == 0) a.pd[(int64_t) (mb >> 5)*a.T + t] = round_f16(d);
            }
        }
    }
}

// ---- the kernel --------------------------------------------------------------------------------------------------------------------
// SMF: the tile of scale products dw[m]*dx[t] comes from a rank-1 f16 MFMA (Q4_0 / Q5_0 / Q8_0 only)
template <int WT, int BMT, int BNT, bool SMF>
__device__ __forceinline__ void mmq_tile(const MmqArgs & a, const int tile, char * lds) {
    constexpr bool Q4K = WT == MI355X_TYPE_Q4_K;
    static_assert(!(Q4K && SMF), "Q8_K scales are not f16 values");
    constexpr int MT = BMT / 64, NT = BNT / 64;               // 32 x 32 tiles per wave
    constexpr int PA = BMT * 4 / 256;                         // (row, block) pairs of A per thread and K-step: 2 or 1
    constexpr int TPR = 4 / PA;                               // threads per A row
    constexpr int CB = BNT / 32;                              // 16-byte chunks of B per thread and K-step
    static_assert(PA == 1 || PA == 2, "BMT is 64 or 128");
    // one stage: A tile | B tile | sA[4][BMT] | mA[4][BMT] (Q4_K) | sB[4][BNT] | dB[BNT] (Q4_K)
    constexpr int OFF_B  = BMT * 128;
    constexpr int OFF_SA = OFF_B + BNT * 128;
    constexpr int OFF_MA = OFF_SA + 4 * BMT * 4;
    constexpr int OFF_SB = OFF_MA + (Q4K ? 4 * BMT * 4 : 0);
    constexpr int OFF_DB = OFF_SB + 4 * BNT * 4;
    // SMF: a block of zeros the size of the larger scale array — what lanes 32..63 (K-slots 8..15 of the rank-1 scale MFMA) read, at
    // the same constant offsets lanes 0..31 use into sA / sB (so that the per-block offset can be an instruction immediate for every lane)
    constexpr int ZB     = SMF ? 4 * (BMT > BNT ? BMT : BNT) * 4 : 16;
    constexpr int OFF_Z  = OFF_DB + (Q4K ? BNT * 4 : 0);
    constexpr int STAGE  = OFF_Z + ZB;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int mi = a.m_major ? tile / a.nt : tile % a.mt;
    const int ni = a.m_major ? tile % a.nt : tile / a.mt;
    const int m0 = mi * BMT;
    const int64_t n0 = (int64_t) ni * BNT;
    const int nk = a.K / MQ_KS;
    const int nb = a.K >> 5;

    // staging assignments (rows past the matrix edge are clamped: their results are never stored)
    //   A (Q4_0 / Q5_0 / Q8_0): pair j = tile row (tid >> 2) + 64*j, block tid & 3;  A (Q4_K): tile row tid / TPR, sub-blocks (tid % TPR) * PA ..
    //   B: 16-byte chunk i of the thread = tile row (tid >> 3) + 32*i, slot tid & 7: eight lanes move one whole 128-byte line
    //   B scales: value i of the thread = element tid + 256*i of the [4][BNT] scale block: a wave reads 256 contiguous bytes
    const int arow = Q4K ? tid / TPR : tid >> 2, alb0 = Q4K ? (tid % TPR) * PA : tid & 3;
    const int arow_g0 = m0 + arow < a.M ? m0 + arow : a.M - 1;
    const int arow_g1 = m0 + arow + 64 < a.M ? m0 + arow + 64 : a.M - 1;
    const int brow = tid >> 3, bslot = tid & 7;
    // running pointers: every operand advances by a constant number of bytes per K-step (no 64-bit multiplies inside the loop)
    const int8_t * bq[CB];
    #pragma unroll
    for (int i = 0; i < CB; i++) {
        const int64_t r = n0 + brow + 32*i;
        bq[i] = a.Bq + (r < a.T ? r : a.T - 1) * a.K + bslot * 16;
    }
    constexpr int SBN = BNT * 4 / 256;                        // B scales per thread and K-step (1 or 2)
    const float * bdp[SBN]; const int * bsp[SBN]; const float * bd8p[SBN];
    #pragma unroll
    for (int i = 0; i < SBN; i++) {
        const int e = tid + 256*i;
        const int64_t c = n0 + e % BNT;
        const int64_t cc = c < a.T ? c : a.T - 1;
        bdp[i]  = a.Bd + (int64_t) (e / BNT) * a.T + cc;       // Q8_0 rows: scale of block e / BNT of the K-step
        bsp[i]  = Q4K ? a.Bs + (int64_t) (e / BNT) * a.T + cc : nullptr;
        bd8p[i] = a.Bd + cc;                                    // Q8_K rows: the super-block's scale
    }
    const int64_t step_scales = 4 * a.T;                      // elements per K-step in a block-major scale array
    // A: byte addresses of this thread's first block in each plane
    const char * aq0, * aq1 = nullptr; const char * ah0 = nullptr, * ah1 = nullptr; const char * ad0 = nullptr, * ad1 = nullptr;
    int a_sb0 = 0;
    {
        const int nbk = a.K >> 5;
        if constexpr (Q4K) {
            const qplanes<MI355X_TYPE_Q4_K> p(a.A, a.nbt);
            a_sb0 = arow_g0 * (a.K >> 8);
            aq0 = (const char *) p.qs + (int64_t) a_sb0 * 128 + (alb0 >> 1) * 32;
        } else {
            const qplanes<WT> p(a.A, a.nbt);
            constexpr int QB = WT == MI355X_TYPE_Q8_0 ? 32 : 16;
            const int64_t ib0 = (int64_t) arow_g0 * nbk + alb0, ib1 = (int64_t) arow_g1 * nbk + alb0;
            aq0 = (const char *) p.qs + ib0 * QB; aq1 = (const char *) p.qs + ib1 * QB;
            ad0 = (const char *) (p.d + ib0);     ad1 = (const char *) (p.d + ib1);
            if constexpr (WT == MI355X_TYPE_Q5_0) { ah0 = (const char *) (p.qh + ib0); ah1 = (const char *) (p.qh + ib1); }
        }
    }

    mq_aregs<WT, PA> ar;
    uint4 br0, br1, br2 = make_uint4(0, 0, 0, 0), br3 = make_uint4(0, 0, 0, 0);      // (named registers: an array indexed inside the lambdas stays in private memory)
    float bsc0 = 0.0f, bsc1 = 0.0f; float bd80 = 0.0f, bd81 = 0.0f; int bsm0 = 0, bsm1 = 0;
    static_assert(CB == 2 || CB == 4, "B staging is written for 64- and 128-column tiles");

    // loads of K-step kt; called with kt = 0, 1, 2, ... in order (the pointers run along)
    auto load_tile = [&](int kt) {
        if constexpr (Q4K) {
            const qplanes<MI355X_TYPE_Q4_K> p(a.A, a.nbt);
            ar.q0 = *(const uint4 *) aq0; ar.q1 = *(const uint4 *) (aq0 + 16);
            aq0 += 64;                                           // 4 sub-blocks = 2 chunks of 32 bytes
            const int sb = a_sb0 + ((kt*4 + alb0) >> 3);
            ar.dm = p.dm[sb];
            const uint32_t * sc = (const uint32_t *) (p.sc + (int64_t) sb*12);
            ar.sc[0] = sc[0]; ar.sc[1] = sc[1]; ar.sc[2] = sc[2];
        } else {
            constexpr int QB = WT == MI355X_TYPE_Q8_0 ? 32 : 16;
            if constexpr (WT == MI355X_TYPE_Q8_0) { ar.q0[0] = *(const uint4 *) aq0; ar.q1[0] = *(const uint4 *) (aq0 + 16); }
            else ar.q[0] = *(const uint4 *) aq0;
            if constexpr (WT == MI355X_TYPE_Q5_0) ar.qh[0] = *(const uint32_t *) ah0;
            ar.d[0] = *(const uint16_t *) ad0;
            if constexpr (PA == 2) {
                if constexpr (WT == MI355X_TYPE_Q8_0) { ar.q0[1] = *(const uint4 *) aq1; ar.q1[1] = *(const uint4 *) (aq1 + 16); }
                else ar.q[1] = *(const uint4 *) aq1;
                if constexpr (WT == MI355X_TYPE_Q5_0) ar.qh[1] = *(const uint32_t *) ah1;
                ar.d[1] = *(const uint16_t *) ad1;
            }
            aq0 += 4*QB; aq1 += 4*QB; ad0 += 8; ad1 += 8;
            if constexpr (WT == MI355X_TYPE_Q5_0) { ah0 += 16; ah1 += 16; }
        }
        br0 = *(const uint4 *) bq[0]; br1 = *(const uint4 *) bq[1];
        if constexpr (CB == 4) { br2 = *(const uint4 *) bq[2]; br3 = *(const uint4 *) bq[3]; }
        #pragma unroll
        for (int i = 0; i < CB; i++) bq[i] += MQ_KS;
        if constexpr (Q4K) {
            bd80 = *bd8p[0]; bsm0 = *bsp[0];
            if constexpr (SBN == 2) { bd81 = *bd8p[1]; bsm1 = *bsp[1]; }
            #pragma unroll
            for (int i = 0; i < SBN; i++) { bsp[i] += step_scales; if (kt & 1) bd8p[i] += a.T; }
        } else {
            bsc0 = *bdp[0];
            if constexpr (SBN == 2) bsc1 = *bdp[1];
            #pragma unroll
            for (int i = 0; i < SBN; i++) bdp[i] += step_scales;
        }
    };
    auto store_tile = [&](int kt, char * st) {
        mq_a_store<WT, PA, BMT, SMF>(ar, st, (float *) (st + OFF_SA), (float *) (st + OFF_MA), arow, alb0, kt*4 + alb0, (const uint32_t *) (lds + 2*STAGE));
        *(uint4 *) (st + OFF_B + mq_off(brow, bslot))      = br0;
        *(uint4 *) (st + OFF_B + mq_off(brow + 32, bslot)) = br1;
        if constexpr (CB == 4) {
            *(uint4 *) (st + OFF_B + mq_off(brow + 64, bslot)) = br2;
            *(uint4 *) (st + OFF_B + mq_off(brow + 96, bslot)) = br3;
        }
        float * sB = (float *) (st + OFF_SB);
        const int e0 = tid, e1 = tid + 256;                       // element of the [4][BNT] scale block = sblk * BNT + column
        if constexpr (Q4K) {
            sB[e0] = bd80 * (float) bsm0;
            if constexpr (SBN == 2) sB[e1] = bd81 * (float) bsm1;
            if (e0 < BNT) ((float *) (st + OFF_DB))[e0] = bd80;      // (block 0's threads hold every column's d8 once)
        } else if constexpr (SMF) {
            ((uint32_t *) sB)[e0] = (uint32_t) f2h(bsc0);                        // exact: a Q8_0 scale is an f16 value
            if constexpr (SBN == 2) ((uint32_t *) sB)[e1] = (uint32_t) f2h(bsc1);
        } else {
            sB[e0] = bsc0;
            if constexpr (SBN == 2) sB[e1] = bsc1;
        }
    };

    floatx16 acc[MT][NT];
    #pragma unroll
    for (int i = 0; i < MT; i++)
        #pragma unroll
        for (int j = 0; j < NT; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    for (int e = tid; e < 2 * (ZB / 4); e += 256) ((uint32_t *) (lds + (e / (ZB / 4)) * STAGE + OFF_Z))[e % (ZB / 4)] = 0u;
    if (tid < 16) ((uint32_t *) (lds + 2*STAGE))[tid] = ((tid & 1) ? 0xF0u : 0u) | ((tid & 2) ? 0xF000u : 0u) | ((tid & 4) ? 0xF00000u : 0u) | ((tid & 8) ? 0xF0000000u : 0u);
    load_tile(0);
    if constexpr (WT == MI355X_TYPE_Q5_0) __syncthreads();      // the table must be complete before ANY thread unpacks with it (round 4: without this barrier
                                                                 // the first tile raced with the 16 threads that write it — wrong bytes only under load)
    store_tile(0, lds);
    __syncthreads();

    const int l31 = lane & 31, hf = lane >> 5;
    // C operand of every integer MFMA: 0x4B400000 = bits(12582912.0f) in all 16 elements, so that D read as a float is 12582912 + sum
    // (from an asm: left to itself the compiler re-materialises the 16 registers in front of every MFMA)
    int mg; asm volatile("v_mov_b32 %0, 0x4b400000" : "=v"(mg));
    const i32x16_t zi = { mg, mg, mg, mg, mg, mg, mg, mg, mg, mg, mg, mg, mg, mg, mg, mg };
    const floatx16 zf = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    typedef float float2_t __attribute__((ext_vector_type(2)));
    const float2_t cmg = { 12582912.0f, 12582912.0f };

    // Compute schedule of one K-step.  Left alone, hipcc issues all 32 MFMAs of a step first, sinks the (memory-free) cvt / fma chains of
    // ALL tile-blocks below the last of them and keeps 16 result sets (512 registers) alive; sched_barrier alone only pins what the IR
    // passes left.  So the order is carried as DATA dependencies by empty asm statements ("+v" operands) and then pinned:
    //   issue(n + 1)  — the MFMA pair of the next tile-block; its asm also names tile-block n's results, whose fold can therefore not
    //                   start before these MFMAs are in the pipe
    //   fold(n)       — 16 cvt + 16 fma in their shadow; ends with the accumulator tile named, so it cannot sink below the next issue
    // Two result sets in flight (64 registers).  A block's fragments and scale operands are requested from LDS right after the LAST
    // issue of the previous block (its two remaining folds cover the latency) into the SAME registers: nothing is double-buffered.
    // The scale operands are 4-register tuples of which only dword 0 is ever written; their zeros come from an asm (opaque to constant
    // propagation) so that the tuples stay loop-carried registers instead of being re-assembled with three v_mov per use.
    struct frags { i32x4_t af[MT], bf[NT]; i32x4_t as[SMF ? MT : 1], bs[SMF ? NT : 1]; float sa[SMF ? 1 : MT][SMF ? 1 : 16], ma[Q4K ? MT : 1][Q4K ? 16 : 1], dxn[SMF ? 1 : NT], vbn[Q4K ? NT : 1]; };
    frags f;
    if constexpr (SMF) {
        int z0; asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
        #pragma unroll
        for (int i = 0; i < MT; i++) f.as[i] = i32x4_t{ z0, z0, z0, z0 };
        #pragma unroll
        for (int j = 0; j < NT; j++) f.bs[j] = i32x4_t{ z0, z0, z0, z0 };
    }
    // LDS addresses of this lane's fragment reads: the XOR swizzle of a row depends on (row >> 1) & 7 only, i.e. on the lane (tile rows
    // start at multiples of 32), so the four blocks' slot offsets are four per-lane constants and everything else — tile index, array
    // base, block of the scale arrays — is an instruction immediate: two address adds per block instead of eleven
    int xo[4];
    #pragma unroll
    for (int b = 0; b < 4; b++) xo[b] = ((2*b + hf) ^ ((l31 >> 1) & 7)) << 4;
    const int a_lane = (wm*(MT*32) + l31) * 128, b_lane = OFF_B + (wn*(NT*32) + l31) * 128;
    const int sa_lane = hf == 0 ? OFF_SA + (wm*(MT*32) + l31) * 4 : OFF_Z + l31 * 4;
    const int sb_lane = hf == 0 ? OFF_SB + (wn*(NT*32) + l31) * 4 : OFF_Z + l31 * 4;
    auto read_frags = [&](const char * st, int b) {
        const float * sA = (const float *) (st + OFF_SA), * mA = (const float *) (st + OFF_MA), * sB = (const float *) (st + OFF_SB);
        const char * pa = st + a_lane + xo[b], * pb = st + b_lane + xo[b];
        #pragma unroll
        for (int i = 0; i < MT; i++) f.af[i] = *(const i32x4_t *) (pa + i*4096);
        #pragma unroll
        for (int j = 0; j < NT; j++) f.bf[j] = *(const i32x4_t *) (pb + j*4096);
        if constexpr (SMF) {
            // rank-1 scale tiles: K-slot 0 of the A / B operand = element 0 of lanes 0..31; lanes 32..63 read the stage's zero block
            #pragma unroll
            for (int i = 0; i < MT; i++) f.as[i][0] = *(const int *) (st + sa_lane + (b*BMT + i*32) * 4);
            #pragma unroll
            for (int j = 0; j < NT; j++) f.bs[j][0] = *(const int *) (st + sb_lane + (b*BNT + j*32) * 4);
        } else {
            // scales of this lane's 16 rows per tile: four float4 broadcast reads (rows 8g + 4*hf .. + 3); its column's scale is a scalar
            #pragma unroll
            for (int j = 0; j < NT; j++) {
                const int c = wn*(NT*32) + j*32 + l31;
                if constexpr (Q4K) { f.dxn[j] = ((const float *) (st + OFF_DB))[c]; f.vbn[j] = sB[b*BNT + c]; }
                else f.dxn[j] = sB[b*BNT + c];
            }
            #pragma unroll
            for (int i = 0; i < MT; i++)
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int row = wm*(MT*32) + i*32 + 8*g + 4*hf;
                    const float4 s4 = *(const float4 *) (sA + b*BMT + row);
                    f.sa[i][4*g] = s4.x; f.sa[i][4*g+1] = s4.y; f.sa[i][4*g+2] = s4.z; f.sa[i][4*g+3] = s4.w;
                    if constexpr (Q4K) { const float4 m4 = *(const float4 *) (mA + b*BMT + row); f.ma[i][4*g] = m4.x; f.ma[i][4*g+1] = m4.y; f.ma[i][4*g+2] = m4.z; f.ma[i][4*g+3] = m4.w; }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // tile-block n of a block = (i, j) = (n % MT, n / MT)
    constexpr int NTB = MT * NT;
    static_assert(NTB == 2 || NTB == 4, "2 or 4 tile-blocks per wave and block");
    auto issue = [&](int n, i32x16_t & S, floatx16 & SC, i32x16_t & Sp, floatx16 & SCp, bool tie) {
        const int i = n % MT, j = n / MT;
        S = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.af[i], f.bf[j], zi, 0, 0, 0);
        if constexpr (SMF) {
            SC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, f.as[i]), __builtin_bit_cast(half8_t, f.bs[j]), zf, 0, 0, 0);
            if (tie) asm volatile("" : "+v"(S), "+v"(SC), "+v"(Sp), "+v"(SCp));
            else     asm volatile("" : "+v"(S), "+v"(SC));
        } else {
            if (tie) asm volatile("" : "+v"(S), "+v"(Sp));
            else     asm volatile("" : "+v"(S));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // scale operands of the VALU form are copied out before the next block's reads overwrite them
    auto fold = [&](int n, const i32x16_t & S, const floatx16 & SC) {
        const int i = n % MT, j = n / MT;
        // float(sum) = (12582912 + sum) - 12582912, exact; two accumulator elements per VALU instruction
        const floatx16 Sf = __builtin_bit_cast(floatx16, S);
        #pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float2_t t = float2_t{ Sf[r], Sf[r + 1] } - cmg;
            float2_t av = { acc[i][j][r], acc[i][j][r + 1] };
            if constexpr (SMF) av = __builtin_elementwise_fma(t, float2_t{ SC[r], SC[r + 1] }, av);
            else {
                av = __builtin_elementwise_fma(t, float2_t{ f.sa[i][r], f.sa[i][r + 1] } * float2_t{ f.dxn[j], f.dxn[j] }, av);
                if constexpr (Q4K) av = __builtin_elementwise_fma(float2_t{ f.ma[i][r], f.ma[i][r + 1] }, float2_t{ f.vbn[j], f.vbn[j] }, av);
            }
            acc[i][j][r] = av[0]; acc[i][j][r + 1] = av[1];
        }
        asm volatile("" : "+v"(acc[i][j]));
        __builtin_amdgcn_sched_barrier(0);
    };
    // one block; nb_ >= 0: the next block's fragments are requested as soon as this block's last MFMAs are issued (SMF), or — the VALU
    // form still needs this block's scale registers in its folds — after its last fold
    auto block = [&](const char * st, int nb_) {
        i32x16_t S0, S1; floatx16 SC0 = zf, SC1 = zf;
        issue(0, S0, SC0, S1, SC1, false);
        issue(1, S1, SC1, S0, SC0, true);
        if constexpr (NTB == 4) {
            fold(0, S0, SC0);
            issue(2, S0, SC0, S1, SC1, true);
            fold(1, S1, SC1);
            issue(3, S1, SC1, S0, SC0, true);
            if constexpr (SMF) { if (nb_ >= 0) read_frags(st, nb_); }
            fold(2, S0, SC0);
            fold(3, S1, SC1);
        } else {
            if constexpr (SMF) { if (nb_ >= 0) read_frags(st, nb_); }
            fold(0, S0, SC0);
            fold(1, S1, SC1);
        }
        if constexpr (!SMF) { if (nb_ >= 0) read_frags(st, nb_); }
    };

    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) load_tile(kt + 1);
        const char * st = lds + (kt & 1) * STAGE;
        __builtin_amdgcn_sched_barrier(0);
        read_frags(st, 0);
        block(st, 1);
        block(st, 2);
        block(st, 3);
        block(st, -1);
        if (kt + 1 < nk) store_tile(kt + 1, lds + ((kt + 1) & 1) * STAGE);
        __syncthreads();
    }
    mq_epilogue<MT, NT>(a, acc, m0 + wm*(MT*32), n0 + wn*(NT*32), lane);
}

template <int WT, int BMT, int BNT, bool SMF>
__global__ void __launch_bounds__(256, 2) k_mmq(const MmqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char mq_lds[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = xcd * a.per + idx;
    if (tile >= a.mt * a.nt) return;                  // padding blocks of the last XCD range (uniform exit, before any barrier)
    mmq_tile<WT, BMT, BNT, SMF>(a, tile, mq_lds);
}

// Grouped form: up to MMQ_GROUP_MAX independent products on the SAME activation rows and of the same shape (the Q / K / V projections
// of an encoder layer; the cross-attention K / V projections of consecutive decoder layers), one launch over the concatenated tile
// space, dealt to the XCDs in contiguous ranges like the single form (gemm_mfma.hip: k_gemm_f16_ring_group).  Every tile runs exactly the
// code of the single form: bit-identical.
#define MMQ_GROUP_MAX 8
struct MmqGroupArgs { MmqArgs g[MMQ_GROUP_MAX]; int n, tiles_per_member, per; };
template <int WT, int BMT, int BNT, bool SMF>
__global__ void __launch_bounds__(256, 2) k_mmq_group(const MmqGroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char mq_lds_g[];
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t = xcd * ga.per + idx;
    if (t >= ga.n * ga.tiles_per_member) return;
    const int gi = t / ga.tiles_per_member;
    mmq_tile<WT, BMT, BNT, SMF>(ga.g[gi], t - gi * ga.tiles_per_member, mq_lds_g);
}

template <int WT, int BMT, int BNT>
static constexpr uint32_t mmq_lds_bytes() {
    constexpr bool Q4K = WT == MI355X_TYPE_Q4_K;
    // two stages (tiles | scales | zero block, SMF or not: the larger) + the Q5_0 unpack table
    return 2u * (uint32_t) ((BMT + BNT) * 128 + 4 * BMT * 4 * (Q4K ? 2 : 1) + 4 * BNT * 4 + (Q4K ? BNT * 4 : 0) + 4 * (BMT > BNT ? BMT : BNT) * 4) + 64u;
}

// tile counts of one member + the XCD-aware tile order for ranges of `per` consecutive tiles (gemm_mfma.hip: ring_tiling)
template <int BMT, int BNT>
static void mmq_tiling(MmqArgs & k, int64_t per) {
    k.mt = (k.M + BMT - 1) / BMT; k.nt = (int) ((k.T + BNT - 1) / BNT);
    const int64_t ntiles = (int64_t) k.mt * k.nt;
    k.per = (int) (per < ntiles ? per : ntiles);
    const double a_tile = (double) BMT * k.K * 0.7, b_tile = (double) BNT * k.K * 1.125;
    const double col_major = a_tile * (k.per < k.mt ? k.per : k.mt) + b_tile * ((k.per + k.mt - 1) / k.mt + (k.per % k.mt ? 1 : 0));
    const double row_major = b_tile * (k.per < k.nt ? k.per : k.nt) + a_tile * ((k.per + k.nt - 1) / k.nt + (k.per % k.nt ? 1 : 0));
    k.m_major = row_major < col_major ? 1 : 0;
}

template <typename F>
static int mmq_lds_attr(mi355x_ctx * ctx, F func, uint32_t lds, std::atomic<bool> * attr_set) {
    const int dev = ctx->device & 63;
    if (lds > 64 * 1024 && !attr_set[dev].load()) {
        if (hipFuncSetAttribute((const void *) func, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) { (void) hipGetLastError(); return MI355X_E_UNSUPPORTED; }
        attr_set[dev].store(true);
    }
    return 0;
}

template <int WT, int BMT, int BNT, bool SMF>
static int launch_mmq(mi355x_ctx * ctx, const MmqArgs * members, int n, double bytes, double flops) {
    constexpr uint32_t lds = mmq_lds_bytes<WT, BMT, BNT>();
    const int64_t tiles = (int64_t) ((members[0].M + BMT - 1) / BMT) * ((members[0].T + BNT - 1) / BNT);
    const int64_t per = (n * tiles + 7) / 8;
    if (n == 1) {
        MmqArgs k = members[0];
        mmq_tiling<BMT, BNT>(k, per);
        k.per = (int) per;
        static std::atomic<bool> attr_set[64];
        if (mmq_lds_attr(ctx, k_mmq<WT, BMT, BNT, SMF>, lds, attr_set) != 0) return MI355X_E_UNSUPPORTED;
        return emit(ctx, "mmq", k_mmq<WT, BMT, BNT, SMF>, dim3((uint32_t) (8 * per)), dim3(256), lds, k, bytes, flops);
    }
    MmqGroupArgs ga; memset(&ga, 0, sizeof(ga));
    for (int i = 0; i < n; i++) { ga.g[i] = members[i]; mmq_tiling<BMT, BNT>(ga.g[i], per); }
    ga.n = n; ga.tiles_per_member = (int) tiles; ga.per = (int) per;
    static std::atomic<bool> attr_set_g[64];
    if (mmq_lds_attr(ctx, k_mmq_group<WT, BMT, BNT, SMF>, lds, attr_set_g) != 0) return MI355X_E_UNSUPPORTED;
    return emit(ctx, "mmq_group", k_mmq_group<WT, BMT, BNT, SMF>, dim3((uint32_t) (8 * per)), dim3(256), lds, ga, bytes, flops);
}

// tile shape: 64 x 128 — the A unpack is amortised over the tile's 128 columns, the B tile is a plain copy, and two result sets, the
// constant C operand and the prefetch registers fit 256 VGPRs beside 2 accumulator tiles per wave (with 4 — a 128 x 128 tile — they
// spill); test option MI355X_OPT_MMQ_TILE = 12864 selects 128 x 64 (bit-identical, tests/test_gpu_mmq.py)
template <int WT, bool SMF>
static int launch_mmq_shape(mi355x_ctx * ctx, const MmqArgs * members, int n, double bytes, double flops) {
    if constexpr (SMF) {
        const int force = mi355x_opt(MI355X_OPT_MMQ_TILE, 0);
        if (force == 12864) return launch_mmq<WT, 128, 64, true>(ctx, members, n, bytes, flops);
    }
    return launch_mmq<WT, 64, 128, SMF>(ctx, members, n, bytes, flops);
}

static int launch_mmq_any(mi355x_ctx * ctx, int wt, const MmqArgs * members, int n, double bytes, double flops) {
    // test option MI355X_OPT_MMQ_SCALE_MFMA = 0: scale products on the VALU from broadcast LDS reads (the Q4_K form) instead of the rank-1 MFMA
    const bool smf = mi355x_opt(MI355X_OPT_MMQ_SCALE_MFMA, 1) != 0;
    switch (wt) {
        case MI355X_TYPE_Q4_0: return smf ? launch_mmq_shape<MI355X_TYPE_Q4_0, true>(ctx, members, n, bytes, flops) : launch_mmq_shape<MI355X_TYPE_Q4_0, false>(ctx, members, n, bytes, flops);
        case MI355X_TYPE_Q5_0: return smf ? launch_mmq_shape<MI355X_TYPE_Q5_0, true>(ctx, members, n, bytes, flops) : launch_mmq_shape<MI355X_TYPE_Q5_0, false>(ctx, members, n, bytes, flops);
        case MI355X_TYPE_Q8_0: return smf ? launch_mmq_shape<MI355X_TYPE_Q8_0, true>(ctx, members, n, bytes, flops) : launch_mmq_shape<MI355X_TYPE_Q8_0, false>(ctx, members, n, bytes, flops);
        default:               return launch_mmq_shape<MI355X_TYPE_Q4_K, false>(ctx, members, n, bytes, flops);
    }
}

// ---- held-back products (mi355x_ctx::pending_*): independent neighbours on the same activation rows leave as ONE grouped launch ----
struct PendingMmq { MmqArgs k[MMQ_GROUP_MAX]; int wt; double bytes, flops; };
static_assert(sizeof(PendingMmq) <= sizeof(((mi355x_ctx *) nullptr)->pending_store), "pending_store too small");

static bool mq_overlap(const void * a, int64_t na, const void * b, int64_t nb) {
    return a && b && na > 0 && nb > 0 && (const char *) a < (const char *) b + nb && (const char *) b < (const char *) a + na;
}
static int64_t mq_dst_bytes(const MmqArgs & k) { return (k.T - 1) * k.dst_nb1 + (int64_t) k.M * (k.dst_f16 ? 2 : 4); }
static int64_t mq_res_bytes(const MmqArgs & k) { return k.residual ? (k.T - 1) * k.res_nb1 + (int64_t) k.M * 4 : 0; }

// may `k` join the held-back members?  Same rows and shape, and no member reads or writes what another member writes.
static bool mmq_mergeable(const PendingMmq & P, int n, int wt, const MmqArgs & k) {
    const MmqArgs & f = P.k[0];
    if (k.pq || f.pq || wt != P.wt) return false;                          // a product that also writes activation rows goes alone
    if (n >= MMQ_GROUP_MAX || k.Bq != f.Bq || k.K != f.K || k.T != f.T || k.M != f.M) return false;
    const int64_t kd = mq_dst_bytes(k);
    if (mq_overlap(k.dst, kd, k.Bq, k.T * (int64_t) k.K * 2)) return false;
    for (int i = 0; i < n; i++) {
        const MmqArgs & e = P.k[i];
        const int64_t ed = mq_dst_bytes(e);
        if (mq_overlap(k.dst, kd, e.dst, ed)) return false;
        if (mq_overlap(k.residual, mq_res_bytes(k), e.dst, ed) || mq_overlap(e.residual, mq_res_bytes(e), k.dst, kd)) return false;
        if (mq_overlap(k.bias, (k.bias_t ? k.T : (int64_t) k.M) * 4, e.dst, ed) || mq_overlap(e.bias, (e.bias_t ? e.T : (int64_t) e.M) * 4, k.dst, kd)) return false;
    }
    return true;
}

static int flush_pending_mmq(mi355x_ctx * ctx) {
    PendingMmq P; memcpy(&P, ctx->pending_store, sizeof(P));
    const int n = ctx->pending_n;
    ctx->pending_n = 0;
    ctx->in_flush = true;
    const int rc = launch_mmq_any(ctx, P.wt, P.k, n, P.bytes, P.flops);
    ctx->in_flush = false;
    return rc;
}

static int hold_mmq(mi355x_ctx * ctx, int wt, const MmqArgs & k, double bytes, double flops) {
    const bool group_on = mi355x_opt(MI355X_OPT_MMQ_GROUP, 1) != 0;      // (test option: single launches, compared with the grouped form in one process)
    if (!group_on) return launch_mmq_any(ctx, wt, &k, 1, bytes, flops);
    PendingMmq * P = (PendingMmq *) ctx->pending_store;
    if (ctx->pending_n > 0 && (ctx->pending_flush != flush_pending_mmq || !mmq_mergeable(*P, ctx->pending_n, wt, k))) {
        const int rc = mi355x_flush_pending(ctx);
        if (rc) return rc;
    }
    if (ctx->pending_n == 0) { P->bytes = 0; P->flops = 0; P->wt = wt; }
    P->k[ctx->pending_n++] = k;
    P->bytes += bytes; P->flops += flops;
    ctx->pending_flush = flush_pending_mmq;
    return 0;
}

// mi355x_gemm_q8act (mi355x_kernels.h): w planar quantized [K, M]; act = activation rows of x [K, T]
static int gemm_q8act_impl(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act, int64_t T, void * dst, int64_t dst_nb1, int dst_type,
                           const mi355x_epilogue * ep, void * prep_out, int prep_only) {
    if (dst_type != MI355X_TYPE_F32 && dst_type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
    const int K = (int) A->ne[0], M = (int) A->ne[1];
    const int wt = A->type;
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0 && wt != MI355X_TYPE_Q4_K) return MI355X_E_UNSUPPORTED;
    if (!t_is_contiguous(A) || A->ne[2] != 1 || A->ne[3] != 1 || M <= 0 || T <= 0 || K <= 0 || K % MQ_KS || (wt == MI355X_TYPE_Q4_K && K % 256)) return MI355X_E_UNSUPPORTED;
    if (((uintptr_t) A->data % 16) || ((uintptr_t) act % 16) || !dst) return MI355X_E_UNSUPPORTED;
    const bool q8k = wt == MI355X_TYPE_Q4_K;
    const qrows_t B = qrows_of((void *) act, q8k, K, T);
    MmqArgs k; memset(&k, 0, sizeof(k));
    k.A = (const char *) A->data; k.nbt = (int64_t) M * (K / type_block(wt));
    k.Bq = B.q; k.Bd = B.d; k.Bs = B.bsum; k.M = M; k.K = K; k.T = T;
    k.dst = (char *) dst; k.dst_nb1 = dst_nb1; k.dst_f16 = dst_type == MI355X_TYPE_F16; k.gelu_tab = ctx->gelu_tab;
    if (ep) { k.bias_t = ep->bias && ep->bias_per_col; k.bias = ep->bias; k.scale = ep->scale; k.has_scale = ep->has_scale; k.gelu = ep->gelu; k.residual = (const char *) ep->residual; k.res_nb1 = ep->residual_nb1; }
    if (prep_out) {
        if (M % MQ_KS || ((uintptr_t) prep_out % 16)) return MI355X_E_UNSUPPORTED;
        const qrows_t P = qrows_of(prep_out, 0, M, T);
        k.pq = P.q; k.pd = P.d; k.prep_only = prep_only;
    }
    const double flops = 2.0 * M * (double) K * (double) T;
    const double bytes = (double) mi355x_type_row_bytes(wt, K) * M + (double) qrows_bytes(q8k, K, T) + (prep_only ? 0.0 : (double) T*M*(k.dst_f16 ? 2 : 4)) + (prep_out ? (double) qrows_bytes(0, M, T) : 0.0);
    return hold_mmq(ctx, wt, k, bytes, flops);       // leaves with the next flush (any other launch, synchronize, end of the graph range)
}

extern "C" size_t mi355x_act_rows_bytes(int wtype, int64_t K, int64_t T) { return (qrows_bytes(wtype == MI355X_TYPE_Q4_K, K, T) + 15) & ~(size_t) 15; }

extern "C" int mi355x_gemm_q8act(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act_rows, int64_t T, void * dst, int64_t dst_nb1, int dst_type,
                                 const mi355x_epilogue * ep) {
    return gemm_q8act_impl(ctx, A, act_rows, T, dst, dst_nb1, dst_type, ep, nullptr, 0);
}

extern "C" int mi355x_gemm_q8act_prep(mi355x_ctx * ctx, const mi355x_tensor * A, const void * act_rows, int64_t T, void * dst, int64_t dst_nb1,
                                      const mi355x_epilogue * ep, void * prep_rows_out) {
    if (!prep_rows_out) return MI355X_E_UNSUPPORTED;
    if (dst && (((uintptr_t) dst % 16) || (dst_nb1 % 16))) return MI355X_E_UNSUPPORTED;
    return gemm_q8act_impl(ctx, A, act_rows, T, dst ? dst : prep_rows_out, dst ? dst_nb1 : A->ne[1]*4, MI355X_TYPE_F32, ep, prep_rows_out, dst ? 0 : 1);
}
