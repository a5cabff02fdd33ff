// Decoder mat-vecs of wide cross-state batches (9..32 columns) on the matrix cores.
//
// k_gemv_q (decode_q.hip) multiplies every weight block with every column on the VALU: 8 v_dot4_i32_i8 + 4 LDS reads + fix-up per
// (block, column), and every one of its 256-1024 workgroups pulls the whole activation image (T*K*1.25 bytes) through LDS — at 16
// columns a 1280 x 1280 product costs 7.2 us, fc1 13.4 us, growing linearly with the columns (profiles/r04_trace16_kernel_stats.csv).
// Here one v_mfma_i32_16x16x64_i8 forms the integer sums of 16 weight rows x 16 columns for TWO quantization blocks' worth of K
// (64 = 2 x 32), whatever the column count, and the activations never touch LDS:
//
//   * a workgroup owns 16 weight rows (one MFMA tile); its waves split K: wave w takes the blocks k_gemv_q's lanes 8w .. 8w+7 take
//     (lane g of k_gemv_q = blocks g, g+64, g+128 of the row), as four PAIRS of neighbouring blocks (b, b+1);
//   * lane l = 16 kg + i holds for A the 16 int8 of weight row i, K-group kg of the pair (kg 0 / 1 = elements 0..15 / 16..31 of block b,
//     kg 2 / 3 of block b+1), unpacked in registers from the planar Q4_0 / Q5_0 / Q8_0 bytes exactly as it read them from HBM; for B the
//     16 int8 of column i of the same K-group — which is, byte for byte, one uint4 of the lo / hi plane of the activation image
//     (decode_common.h), read straight from L2 into the operand registers;
//   * the MFMA contracts all 64 K; the two blocks' sums must stay apart (each has its own f32 scale), so the pair is issued twice with
//     the other block's half of A masked to zero: S_b, S_b+1;
//   * scale products dw[row] * dx[col] for the tile: one rank-1 v_mfma_f32_16x16x16_f16 per block (dw in K-slot 0 of A, dx — an f16
//     value by construction of Q8_0 — in K-slot 0 of B; exact in f32), as in mmq.hip;
//   * the integer sums arrive as floats (C operand = 0x4B400000, mmq.hip), one subtract each;
//   * f32 accumulation follows k_gemv_q's summation tree STEP FOR STEP — per block p = fma(dw*dx, float(S), chain over the lane's
//     units), then the wave butterfly of k_gemv_q as explicit pairwise adds: (b, b+1) inside the pair, pairs inside the wave, waves
//     through LDS — so a column's value is BIT-IDENTICAL to k_gemv_q's, k_gemv_row's and k_vocab's (tests/test_gpu_batch.py): a state's
//     logits do not depend on how many other states share its chain.
//   * the vocabulary projection uses k_vocab's tree instead (8 leaves of K/256 blocks with stride 8): same kernel, LS = 8.
//
// Reference arithmetic: ggml-cpu/ggml-cpu.c:1322-1357 (src1 -> vec_dot_type), ggml-cpu/quants.c:225-259, :365-406, :451-479 (integer
// block dots, f32 scale-accumulate).  The reference's CUDA backend makes the same switch from mat-vec to tile kernels above 8 columns
// (ggml-cuda/mmvq.cuh:3, mmq.cu:259).
#include "decode_common.h"

typedef int   i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct MXSeg { const void * w; int64_t nbt; const float * bias; float scale; int has_scale, gelu, dst_f16, N; };
struct MXArgs {
    const void * planes; int K, T, nseg, ntot;
    int row_start[4];
    MXSeg seg[3];
    const uint16_t * gelu_tab;
    void * planes_out; int planes_only; int nwt;           // nwt: waves per row tile
    mi355x_gemv_cols cols;
};

#define MX_ROWS 16
#define MX_PSTRIDE 20          // floats per column of a partial tile in LDS (16 rows + 4: the float4 stores of 8 lanes hit 32 different banks)

// One pair's operands as they come from memory
template <int WT, int CG> struct mx_pair {
    u32x4 aq; uint32_t aqh; uint32_t add;          // this lane's 16 weight bytes (Q8_0) / 16 bytes of nibbles, high bits, the pair's two f16 scales
    u32x4 bq[CG]; float2 bdx[CG];                  // this lane's 16 activation bytes per column group, the pair's two activation scales
};

// LS = leaf stride (64: k_gemv_q's tree, 8: k_vocab's), LW = leaves per wave (8 / 2), NU = blocks per leaf, CG = column groups of 16,
// RTP = row tiles per workgroup (2: the 32 rows of one Q8_0 block of the RESULT, whose planes are written as well)
template <int WT, int LS, int LW, int NU, int CG, int RTP, bool NSEG1>
__global__ void __launch_bounds__(RTP * (LS == 8 ? 4 : 8) * 64 > 512 ? 1024 : 512) k_gemv_mx(const MXArgs a) {
    constexpr int NP = LW / 2;                     // pairs per wave and unit
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t lut[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nwt = a.nwt;
    const int wave_all = tid >> 6;
    const int tsel = RTP == 1 ? 0 : __builtin_amdgcn_readfirstlane(wave_all / nwt);
    const int wave = RTP == 1 ? __builtin_amdgcn_readfirstlane(wave_all) : __builtin_amdgcn_readfirstlane(wave_all - tsel * nwt);
    const int i = lane & 15, kg = lane >> 4, blkoff = kg >> 1, half = kg & 1;
    const int K = a.K, nb = K >> 5, T = a.T, ntot = a.ntot;
    const int tile = blockIdx.x * RTP + tsel;
    const int row0_raw = tile * MX_ROWS;
    const bool tile_ok = row0_raw < ntot;
    const int row0 = tile_ok ? row0_raw : 0;
    int s = 0;
    if constexpr (!NSEG1) {
        if (a.nseg > 1 && row0 >= a.row_start[1]) s = 1;
        if (a.nseg > 2 && row0 >= a.row_start[2]) s = 2;
    }
    const MXSeg & sgr = NSEG1 ? a.seg[0] : a.seg[s];
    const int rseg = row0 - (NSEG1 ? 0 : a.row_start[s]);          // first row of the tile inside its segment (segments are multiples of 16 rows)
    const char * wbase = (const char *) sgr.w;
    const int64_t nbt = sgr.nbt;

    if (tid < 16) lut[tid] = ((tid & 1) ? 0xF0u : 0u) | ((tid & 2) ? 0xF000u : 0u) | ((tid & 4) ? 0xF00000u : 0u) | ((tid & 8) ? 0xF0000000u : 0u);

    // ---- per-lane base addresses -------------------------------------------------------------------------------------------------
    // A: weight row rseg + i, first block of the wave's leaves (+ blkoff: this lane's block of a pair)
    const int b0 = LW * wave;                                              // first leaf = first block (unit 0) of the wave
    const int arow = rseg + i < sgr.N ? rseg + i : sgr.N - 1;             // (the last tile of a matrix whose rows are not a multiple of 16: clamped, never stored)
    const int64_t ibrow = (int64_t) arow * nb;
    constexpr int QB = WT == MI355X_TYPE_Q8_0 ? 32 : 16;
    const char * aq_p  = wbase + (ibrow + b0 + blkoff) * QB + (WT == MI355X_TYPE_Q8_0 ? half * 16 : 0);
    const char * aqh_p = wbase + nbt * 16 + (ibrow + b0 + blkoff) * 4;                                   // Q5_0 only
    const char * ad_p  = wbase + nbt * (WT == MI355X_TYPE_Q8_0 ? 32 : (WT == MI355X_TYPE_Q5_0 ? 20 : 16)) + (ibrow + b0) * 2;      // both scales of a pair: one dword
    // B: column i + 16 c of the planes (columns >= T read column T - 1 again; never stored)
    const size_t istride = dg_img_stride(WT, K);
    const char * bq_p[CG]; const char * bdx_p[CG];
    #pragma unroll
    for (int c = 0; c < CG; c++) {
        const int t = i + 16*c, tc = t < T ? t : T - 1;
        const int gi = tc >> 3, ti = tc & 7, Ti = T - 8*gi < 8 ? T - 8*gi : 8;
        const char * img = (const char *) a.planes + (size_t) gi * istride;
        bq_p[c]  = img + (size_t) half * Ti * nb * 16 + ((size_t) ti * nb + b0 + blkoff) * 16;
        bdx_p[c] = img + (size_t) 2 * Ti * nb * 16 + ((size_t) ti * nb + b0) * 4;
    }

    // ---- constants ------------------------------------------------------------------------------------------------------------------
    int mg; asm volatile("v_mov_b32 %0, 0x4b400000" : "=v"(mg));          // C operand of the integer MFMAs: D read as a float = 12582912 + sum
    const i32x4_t cmagic = { mg, mg, mg, mg };
    const f32x4_t zf = { 0.0f, 0.0f, 0.0f, 0.0f };
    const uint32_t m_lo = kg < 2 ? 0xFFFFFFFFu : 0u, m_hi = ~m_lo;        // block b lives in K-groups 0, 1 (lanes 0..31), block b + 1 in 2, 3
    const uint32_t m_s  = kg == 0 ? 0xFFFFFFFFu : 0u;                     // rank-1 scale MFMA: K-slot 0 = element 0 of lanes 0..15
    const int nsh = (WT == MI355X_TYPE_Q8_0) ? 0 : 4 * half;               // this lane's nibble of every byte
    const int hsh = 16 * half;                                             // ... and its 16 high bits (Q5_0)
    typedef int i32x2_t __attribute__((ext_vector_type(2)));

    // ---- loads of unit u: every pair of the wave (clamped: a pair beyond the row reads the row's last pair again and is not used) ----
    mx_pair<WT, CG> pr[2][NP];
    auto load_unit = [&](int u, mx_pair<WT, CG> (&st)[NP]) {
        #pragma unroll
        for (int j = 0; j < NP; j++) {
            int rel = 2*j + LS*u;                                          // block offset from b0
            if (b0 + rel >= nb) rel = nb - 2 - b0;                         // (wave-uniform; nb is even)
            mx_pair<WT, CG> & p = st[j];
            p.aq = __builtin_nontemporal_load((const u32x4 *) (aq_p + (int64_t) rel * QB));
            if constexpr (WT == MI355X_TYPE_Q5_0) p.aqh = *(const uint32_t *) (aqh_p + (int64_t) rel * 4); else p.aqh = 0;
            p.add = *(const uint32_t *) (ad_p + (int64_t) rel * 2);
            #pragma unroll
            for (int c = 0; c < CG; c++) {
                p.bq[c]  = *(const u32x4 *) (bq_p[c] + (int64_t) rel * 16);
                p.bdx[c] = *(const float2 *) (bdx_p[c] + (int64_t) rel * 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    load_unit(0, pr[0]);
    if constexpr (NU > 1) load_unit(1, pr[1]);
    if constexpr (WT == MI355X_TYPE_Q5_0) __syncthreads();                 // the unpack table is complete

    // ---- units -> leaves: accA / accB = the chains of the pair's two leaves (k_gemv_q: one lane's acc over its units) -------------------
    f32x4_t accA[NP][CG], accB[NP][CG];
    #pragma unroll
    for (int j = 0; j < NP; j++)
        #pragma unroll
        for (int c = 0; c < CG; c++) { accA[j][c] = zf; accB[j][c] = zf; }
    #pragma unroll
    for (int u = 0; u < NU; u++) {
        #pragma unroll
        for (int j = 0; j < NP; j++) {
            if (b0 + 2*j + LS*u >= nb) continue;                           // (wave-uniform) the row has no such blocks: the leaf's chain ends
            const mx_pair<WT, CG> & p = pr[u & 1][j];
            // A: this lane's 16 signed bytes
            uint32_t w[4] = { p.aq[0], p.aq[1], p.aq[2], p.aq[3] };
            if constexpr (WT == MI355X_TYPE_Q5_0) {
                const uint32_t inv = (~p.aqh) >> hsh;                      // bit k set: element k of this half is negative (x - 16 = 0xF0 | nib)
                #pragma unroll
                for (int e = 0; e < 4; e++) w[e] = ((w[e] >> nsh) & 0x0F0F0F0Fu) | lut[(inv >> (4*e)) & 0xFu];
            } else if constexpr (WT == MI355X_TYPE_Q4_0) {
                #pragma unroll
                for (int e = 0; e < 4; e++) w[e] = ((((w[e] >> nsh) & 0x0F0F0F0Fu) | 0x80808080u) - 0x08080808u) ^ 0x80808080u;      // nib - 8 per byte (mmq.hip: q4_signed)
            }
            const i32x4_t a_lo = { (int) (w[0] & m_lo), (int) (w[1] & m_lo), (int) (w[2] & m_lo), (int) (w[3] & m_lo) };
            const i32x4_t a_hi = { (int) (w[0] & m_hi), (int) (w[1] & m_hi), (int) (w[2] & m_hi), (int) (w[3] & m_hi) };
            // scale operands: the pair's two f16 weight scales, K-slot 0 only
            const uint32_t dd = p.add & m_s;
            const i32x2_t sa0 = { (int) (dd & 0xFFFFu), 0 }, sa1 = { (int) (dd >> 16), 0 };
            #pragma unroll
            for (int c = 0; c < CG; c++) {
                const i32x4_t bq = { (int) p.bq[c][0], (int) p.bq[c][1], (int) p.bq[c][2], (int) p.bq[c][3] };
                const i32x4_t S0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_lo, bq, cmagic, 0, 0, 0);
                const i32x4_t S1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hi, bq, cmagic, 0, 0, 0);
                const i32x2_t sb0 = { (int) (uint32_t) f2h(p.bdx[c].x), 0 }, sb1 = { (int) (uint32_t) f2h(p.bdx[c].y), 0 };      // exact: a Q8_0 scale is an f16 value
                const f32x4_t SC0 = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4_t, sa0), __builtin_bit_cast(half4_t, sb0), zf, 0, 0, 0);
                const f32x4_t SC1 = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4_t, sa1), __builtin_bit_cast(half4_t, sb1), zf, 0, 0, 0);
                const f32x4_t F0 = __builtin_bit_cast(f32x4_t, S0), F1 = __builtin_bit_cast(f32x4_t, S1);
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    accA[j][c][r] = fmaf(SC0[r], F0[r] - 12582912.0f, accA[j][c][r]);       // k_gemv_q: acc = fmaf(dw*dx, (float) sum, acc)
                    accB[j][c][r] = fmaf(SC1[r], F1[r] - 12582912.0f, accB[j][c][r]);
                }
            }
        }
        if (u + 2 < NU) load_unit(u + 2, pr[u & 1]);                       // the stage just used is refilled two units ahead
    }
    f32x4_t l1[NP][CG];
    #pragma unroll
    for (int j = 0; j < NP; j++)
        #pragma unroll
        for (int c = 0; c < CG; c++) l1[j][c] = accA[j][c] + accB[j][c];   // butterfly step lane ^ 1
    // butterfly steps lane ^ 2, lane ^ 4 inside the wave's leaves.  A leaf without blocks is +0 and x + 0 == x for every x these sums can
    // take (a sum of values that are not -0 is never -0), so adding the empty leaves as k_gemv_q does changes no bit.
    f32x4_t part[CG];
    #pragma unroll
    for (int c = 0; c < CG; c++) {
        if constexpr (NP == 4)      part[c] = (l1[0][c] + l1[1][c]) + (l1[2][c] + l1[3][c]);
        else if constexpr (NP == 2) part[c] = l1[0][c] + l1[1][c];
        else                        part[c] = l1[0][c];
    }
    // ---- partial tiles -> LDS: [tile of the workgroup][wave][column][MX_PSTRIDE], this lane's rows 4 kg .. 4 kg + 3 of column i + 16 c ----
    float * pt = (float *) smem;
    #pragma unroll
    for (int c = 0; c < CG; c++)
        *(f32x4_t *) (pt + ((size_t) (tsel * nwt + wave) * (16*CG) + (i + 16*c)) * MX_PSTRIDE + 4*kg) = part[c];
    __syncthreads();

    // ---- combine over the waves (butterfly steps lane ^ 8, ^ 16, ^ 32), epilogue, stores ----------------------------------------------
    typedef const MXArgs __attribute__((address_space(4))) * kargs_t;
    const kargs_t ka = (kargs_t) __builtin_amdgcn_kernarg_segment_ptr();
    const int nthreads = blockDim.x;
    float * otile = pt + (size_t) RTP * nwt * (16*CG) * MX_PSTRIDE;       // [column][32] result tile of a planes-out workgroup
    for (int o = tid; o < RTP * MX_ROWS * 16 * CG; o += nthreads) {
        const int tl = RTP == 1 ? 0 : o / (MX_ROWS * 16 * CG), e = RTP == 1 ? o : o - tl * (MX_ROWS * 16 * CG);
        const int r = e & 15, t = e >> 4;
        const int trow0 = (blockIdx.x * RTP + tl) * MX_ROWS;
        if (t >= T || trow0 + r >= ntot) continue;
        float P[8];
        #pragma unroll
        for (int w8 = 0; w8 < 8; w8++) P[w8] = w8 < nwt ? pt[((size_t) (tl * nwt + w8) * (16*CG) + t) * MX_PSTRIDE + r] : 0.0f;
        float v = ((P[0] + P[1]) + (P[2] + P[3])) + ((P[4] + P[5]) + (P[6] + P[7]));
        int s2 = 0;
        if constexpr (!NSEG1) {
            if (a.nseg > 1 && trow0 >= a.row_start[1]) s2 = 1;
            if (a.nseg > 2 && trow0 >= a.row_start[2]) s2 = 2;
        }
        const MXSeg & sg = NSEG1 ? a.seg[0] : a.seg[s2];
        const int row = trow0 - (NSEG1 ? 0 : a.row_start[s2]) + r;
        void * dcol = ka->cols.dst[NSEG1 ? 0 : s2][t];
        const float * rcol = ka->cols.res[NSEG1 ? 0 : s2][t];
        if (sg.bias)      v = v + sg.bias[row];
        if (sg.has_scale) v = v * sg.scale;
        if (sg.gelu)      v = gelu_lut(v, a.gelu_tab);
        if (rcol)         v = v + rcol[row];
        if (!(RTP == 2 && a.planes_only)) {
            if (sg.dst_f16) ((uint16_t *) dcol)[row] = f2h(v); else ((float *) dcol)[row] = v;
        }
        if constexpr (LS == 8) { float * mcol = (float *) ka->cols.mirror[t]; if (mcol) mcol[row] = v; }
        if constexpr (RTP == 2) otile[t*32 + tl*MX_ROWS + r] = v;
    }
    if constexpr (RTP == 2) {
        // the workgroup's 32 rows are one Q8_0 block of the result for every column: quantize -> planes of K' = ntot (k_gemv_q POUT)
        __syncthreads();
        if (a.planes_out && tid < 16*CG*8 && (tid >> 3) < T) {
            const int t = tid >> 3, q = tid & 7;
            const float4 x4 = *(const float4 *) (otile + t*32 + q*4);
            const float xv[4] = { x4.x, x4.y, x4.z, x4.w };
            const int nbo = ntot >> 5;
            const int gi = t >> 3, ti = t & 7, Ti = T - 8*gi < 8 ? T - 8*gi : 8;
            uint32_t * olo = (uint32_t *) ((char *) a.planes_out + (size_t) gi * dg_img_stride(MI355X_TYPE_Q8_0, ntot)), * ohi = olo + (size_t) Ti*nbo*4;
            float * odx = (float *) (ohi + (size_t) Ti*nbo*4);
            int * osx = (int *) (odx + Ti*nbo);
            dg_q8_0_store(xv, (int) blockIdx.x*32 + q*4, ti, nbo, olo, ohi, odx, osx);
        }
    }
}

template <int WT, int LS, int LW, int NU, int CG, int RTP, bool NSEG1>
static int mx_emit(mi355x_ctx * ctx, const MXArgs & k, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    return emit(ctx, LS == 8 ? "vocab_mx" : "gemv_mx", k_gemv_mx<WT, LS, LW, NU, CG, RTP, NSEG1>, grid, block, lds, k, bytes, flops);
}

template <int WT, int CG>
static int mx_launch(mi355x_ctx * ctx, const MXArgs & k, bool vocab, int nu, bool pout, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    if (vocab) {
        switch (nu) {
            case 2: return mx_emit<WT, 8, 2, 2, CG, 2, true>(ctx, k, grid, block, lds, bytes, flops);
            case 3: return mx_emit<WT, 8, 2, 3, CG, 2, true>(ctx, k, grid, block, lds, bytes, flops);
            case 4: return mx_emit<WT, 8, 2, 4, CG, 2, true>(ctx, k, grid, block, lds, bytes, flops);
            case 5: return mx_emit<WT, 8, 2, 5, CG, 2, true>(ctx, k, grid, block, lds, bytes, flops);
        }
        return MI355X_E_UNSUPPORTED;
    }
    const bool nseg1 = k.nseg == 1;
    if (pout) {
        if (nu != 1 || !nseg1) return MI355X_E_UNSUPPORTED;
        return mx_emit<WT, 64, 8, 1, CG, 2, true>(ctx, k, grid, block, lds, bytes, flops);
    }
    switch (nu) {
        case 1: return nseg1 ? mx_emit<WT, 64, 8, 1, CG, 1, true>(ctx, k, grid, block, lds, bytes, flops) : mx_emit<WT, 64, 8, 1, CG, 1, false>(ctx, k, grid, block, lds, bytes, flops);
        case 2: return nseg1 ? mx_emit<WT, 64, 8, 2, CG, 1, true>(ctx, k, grid, block, lds, bytes, flops) : MI355X_E_UNSUPPORTED;
        case 3: return nseg1 ? mx_emit<WT, 64, 8, 3, CG, 1, true>(ctx, k, grid, block, lds, bytes, flops) : MI355X_E_UNSUPPORTED;
    }
    return MI355X_E_UNSUPPORTED;
}

// columns from which the matrix-core form is taken (GGML_MI355X_MX_MIN_T; 0 = never).  Below it k_gemv_q / k_vocab run.
static int mx_min_t() {
    static const int v = getenv("GGML_MI355X_MX_MIN_T") ? atoi(getenv("GGML_MI355X_MX_MIN_T")) : 9;
    return v;
}

// mat-vec over prepared activation planes on the matrix cores; MI355X_E_UNSUPPORTED: the caller goes on to k_vocab / k_gemv_q
int mi355x_gemv_mx(mi355x_ctx * ctx, const mi355x_gemv_desc * d) {
    const int min_t = mx_min_t();
    if (min_t <= 0 || d->T < min_t || d->T > MI355X_MAX_COLS) return MI355X_E_UNSUPPORTED;
    if (!d->x_planes || d->x || d->attn_part_o || d->has_norm || ((uintptr_t) d->x_planes % 16)) return MI355X_E_UNSUPPORTED;
    if (d->nseg < 1 || d->nseg > 3) return MI355X_E_UNSUPPORTED;
    const int wt = d->seg[0].wtype, K = d->K, T = d->T;
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0) return MI355X_E_UNSUPPORTED;
    if (K <= 0 || K % 64) return MI355X_E_UNSUPPORTED;                     // pairs of blocks
    const int nb = K / 32;
    MXArgs k; memset(&k, 0, sizeof(k));
    k.planes = d->x_planes; k.K = K; k.T = T; k.nseg = d->nseg; k.gelu_tab = ctx->gelu_tab;
    int ntot = 0; double wbytes = 0;
    for (int s = 0; s < d->nseg; s++) {
        const mi355x_gemv_seg & g = d->seg[s];
        if (g.wtype != wt || g.N <= 0 || ((uintptr_t) g.w % 16)) return MI355X_E_UNSUPPORTED;
        if (g.dst_type != MI355X_TYPE_F32 && g.dst_type != MI355X_TYPE_F16) return MI355X_E_UNSUPPORTED;
        if (g.N % MX_ROWS && s + 1 < d->nseg) return MI355X_E_UNSUPPORTED; // a tile never straddles two segments
        k.row_start[s] = ntot;
        MXSeg & o = k.seg[s];
        o.w = g.w; o.nbt = (int64_t) g.N * nb; o.bias = g.ep.bias; o.scale = g.ep.scale; o.has_scale = g.ep.has_scale; o.gelu = g.ep.gelu; o.dst_f16 = g.dst_type == MI355X_TYPE_F16; o.N = g.N;
        for (int t = 0; t < T; t++) {
            if (d->cols) { k.cols.dst[s][t] = d->cols->dst[s][t]; k.cols.res[s][t] = d->cols->res[s][t]; }
            else {
                k.cols.dst[s][t] = g.dst ? (char *) g.dst + (int64_t) t*g.dst_nb1 : nullptr;
                k.cols.res[s][t] = g.ep.residual ? (const float *) ((const char *) g.ep.residual + (int64_t) t*g.ep.residual_nb1) : nullptr;
            }
            if (!k.cols.dst[s][t] && !(d->planes_out && d->planes_out_only)) return MI355X_E_UNSUPPORTED;
            if ((k.cols.res[s][t] != nullptr) != (k.cols.res[s][0] != nullptr)) return MI355X_E_UNSUPPORTED;
        }
        ntot += g.N;
        wbytes += (double) mi355x_type_row_bytes(wt, K) * g.N;
    }
    for (int s = d->nseg; s < 4; s++) k.row_start[s] = ntot;
    k.ntot = ntot;
    const bool vocab = ntot > 8192;
    bool mirror = false;
    int nu, nwt, rtp = 1;
    const bool pout = d->planes_out != nullptr;
    if (vocab) {
        // k_vocab's shapes and tree (decode_q.hip: mi355x_vocab): 8 lanes per row, lane j = blocks j, j + 8, ...
        const mi355x_gemv_seg & g = d->seg[0];
        if (d->nseg != 1 || pout || K % 256 || K > 2048 || K / 256 < 2 || K / 256 > 5) return MI355X_E_UNSUPPORTED;
        if (g.ep.bias || g.ep.has_scale || g.ep.gelu || g.ep.residual || g.dst_type != MI355X_TYPE_F32) return MI355X_E_UNSUPPORTED;
        mirror = d->cols && d->cols->mirror[0];
        for (int t = 0; t < T; t++) { k.cols.mirror[t] = mirror ? d->cols->mirror[t] : nullptr; if (mirror && !k.cols.mirror[t]) return MI355X_E_UNSUPPORTED; }
        nu = K / 256; nwt = 4; rtp = 2;
    } else {
        nu = (nb + 63) / 64;
        if (nu > 3) return MI355X_E_UNSUPPORTED;
        nwt = ((nb < 64 ? nb : 64) + 7) / 8;
        if (pout) {
            if (d->nseg != 1 || ntot % 32 || nu != 1 || ((uintptr_t) d->planes_out % 16)) return MI355X_E_UNSUPPORTED;
            k.planes_out = d->planes_out; k.planes_only = d->planes_out_only; rtp = 2;
        }
    }
    k.nwt = nwt;
    const int cg = T > 16 ? 2 : 1;
    const int ntiles = (ntot + MX_ROWS - 1) / MX_ROWS;              // (a single segment may end inside its last tile: the vocabulary's 51864 / 51865 / 51866 rows)
    const dim3 grid((ntiles + rtp - 1) / rtp), block(64 * nwt * rtp);
    if (pout && (int) block.x < 16 * cg * 8) return MI355X_E_UNSUPPORTED;   // the planes-out pass needs 8 threads per column
    const uint32_t lds = (uint32_t) ((size_t) rtp * nwt * (16*cg) * MX_PSTRIDE * 4 + (rtp == 2 ? 32 * 16 * cg * 4 : 0));
    const double bytes = wbytes + (double) dg_planes_bytes(wt, K, T) + (double) ntot*T*4;
    const double flops = 2.0 * ntot * K * T;
    int rc;
    #define MX_GO(WT_) (cg == 2 ? mx_launch<WT_, 2>(ctx, k, vocab, nu, pout, grid, block, lds, bytes, flops) : mx_launch<WT_, 1>(ctx, k, vocab, nu, pout, grid, block, lds, bytes, flops))
    switch (wt) {
        case MI355X_TYPE_Q4_0: rc = MX_GO(MI355X_TYPE_Q4_0); break;
        case MI355X_TYPE_Q5_0: rc = MX_GO(MI355X_TYPE_Q5_0); break;
        default:               rc = MX_GO(MI355X_TYPE_Q8_0); break;
    }
    #undef MX_GO
    if (rc == 0 && mirror) ctx->last_mirrored = 1;
    return rc;
}
