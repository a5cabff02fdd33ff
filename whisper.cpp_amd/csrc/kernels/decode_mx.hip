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
#include <atomic>

#define MX_VOCAB_TILES 4        // row tiles (= waves) per workgroup of the vocabulary projection

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

// ---- wave-private staging area of one UNIT (8 neighbouring blocks of the tile's 16 rows and of every column) --------------------------
// A lane of the MFMA needs 16 bytes of ITS row / column: 64 lanes in 16 (rows) or 32 (column planes) different cache lines per load
// instruction, which the texture-address unit serialises (first version of this kernel: 64 cycles per load instruction, fc2 17.7 us —
// profiles/r05_mx_kbench_v1_direct_operand_loads_on.json).  So a unit is fetched the way it LIES in memory — 128-byte runs (8 blocks x 16
// bytes of one row's nibbles, of one column's lo or hi plane), 8 lanes per run — parked in LDS, and read back in operand order.  Row /
// column strides of 144 / 272 / 40 / 20 bytes keep both directions free of bank conflicts (MI355X_MICROARCH.md, LDS table).
template <int WT, int CG> struct mx_lds {
    static constexpr int AQ_ROW = WT == MI355X_TYPE_Q8_0 ? 272 : 144;     // 8 blocks x 32 / 16 bytes + 16
    static constexpr int OFF_AQ  = 0;
    static constexpr int OFF_AQH = OFF_AQ + 16 * AQ_ROW;                  // [16 rows][8 x u32 + 8]      (Q5_0)
    static constexpr int OFF_AD  = OFF_AQH + 16 * 40;                     // [16 rows][8 x f16 + 4]
    static constexpr int OFF_BQ  = OFF_AD + 16 * 20;                      // [2 planes][16 CG columns][8 x 16 + 16]
    static constexpr int BQ_PLANE = 16 * CG * 144;
    static constexpr int OFF_BDX = OFF_BQ + 2 * BQ_PLANE;                 // [16 CG columns][8 x f32 + 8]
    static constexpr int SIZE    = OFF_BDX + 16 * CG * 40;                // multiple of 16
};

// one unit's bytes between the global loads and the LDS stores
template <int WT, int CG> struct mx_unit {
    u32x4 aq[WT == MI355X_TYPE_Q8_0 ? 4 : 2]; uint2 aqh; uint32_t ad;
    u32x4 bq[4 * CG]; uint2 bdx[CG];
};

// LS = leaf stride (64: k_gemv_q's tree, 8: k_vocab's), NU = units per wave (a unit = the 8 neighbouring leaves' blocks u), CG = column
// groups of 16, RTP = row tiles per workgroup (LS 64, RTP 2: the 32 rows of one Q8_0 block of the RESULT, whose planes are written as well)
template <int WT, int LS, int NU, int CG, int RTP, bool NSEG1>
__global__ void __launch_bounds__(RTP * (LS == 8 ? 1 : 8) * 64 > 512 ? 1024 : 512) k_gemv_mx(const MXArgs a) {
    typedef mx_lds<WT, CG> L;
    constexpr bool Q8 = WT == MI355X_TYPE_Q8_0, Q5 = WT == MI355X_TYPE_Q5_0;
    constexpr int NAQ = Q8 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t lut[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nwt = a.nwt;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tsel = RTP == 1 ? 0 : wave_all / nwt;
    const int wave = RTP == 1 ? wave_all : wave_all - tsel * nwt;
    const int i = lane & 15, kg = lane >> 4;
    const int K = a.K, nb = K >> 5, T = a.T, ntot = a.ntot;
    const int tile = blockIdx.x * RTP + tsel;
    const int row0 = tile * MX_ROWS < ntot ? tile * MX_ROWS : 0;
    int s = 0;
    if constexpr (!NSEG1) {
        if (a.nseg > 1 && row0 >= a.row_start[1]) s = 1;
        if (a.nseg > 2 && row0 >= a.row_start[2]) s = 2;
    }
    const MXSeg & sgr = NSEG1 ? a.seg[0] : a.seg[s];                     // (RTP > 1 only with one segment: the workgroup's tiles share it)
    const int seg0 = NSEG1 ? 0 : a.row_start[s];
    const int rseg = row0 - seg0;                                         // first row of the tile inside its segment (segments but the last are multiples of 16 rows)
    const char * wbase = (const char *) sgr.w;
    const int64_t nbt = sgr.nbt;
    const int Nseg = sgr.N;
    char * R = smem + (size_t) wave_all * L::SIZE;                        // this wave's staging area

    if (tid < 16) lut[tid] = ((tid & 1) ? 0xF0u : 0u) | ((tid & 2) ? 0xF000u : 0u) | ((tid & 4) ? 0xF00000u : 0u) | ((tid & 8) ? 0xF0000000u : 0u);

    // ---- the epilogue's operands of this thread's first outputs: requested now, used after the products (two dependent memory round trips
    //      — column pointer, then residual — that would otherwise follow the last barrier) -----------------------------------------------
    typedef const MXArgs __attribute__((address_space(4))) * kargs_t;
    const kargs_t ka = (kargs_t) __builtin_amdgcn_kernarg_segment_ptr();
    const int nthreads = blockDim.x;
    constexpr int NOUT = RTP * MX_ROWS * 16 * CG;                         // outputs of the workgroup: [tile][column][row]
    constexpr int EPI = 2;
    struct epi_t { int row, t, tl; float bias, res; void * dcol; float * mcol; bool ok; };
    auto epi_prep = [&](int o) {
        epi_t e;
        e.tl = RTP == 1 ? 0 : o / (MX_ROWS * 16 * CG);
        const int q = RTP == 1 ? o : o - e.tl * (MX_ROWS * 16 * CG);
        const int r = q & 15;
        e.t = q >> 4;
        const int grow = (blockIdx.x * RTP + e.tl) * MX_ROWS + r;          // row over all segments
        e.ok = o < NOUT && e.t < T && grow < ntot;
        e.row = e.ok ? grow - seg0 : 0;
        const int tc = e.ok ? e.t : 0;
        e.dcol = ka->cols.dst[NSEG1 ? 0 : s][tc];
        const float * rcol = ka->cols.res[NSEG1 ? 0 : s][tc];
        e.mcol = LS == 8 ? (float *) ka->cols.mirror[tc] : nullptr;
        e.bias = sgr.bias ? sgr.bias[e.row] : 0.0f;
        e.res  = rcol ? rcol[e.row] : 0.0f;
        // (has-residual is a property of the segment: every column has one or none — checked by the launcher)
        return e;
    };
    epi_t ep[EPI];
    #pragma unroll
    for (int k = 0; k < EPI; k++) ep[k] = epi_prep(tid + k * nthreads);
    const bool has_res = ka->cols.res[NSEG1 ? 0 : s][0] != nullptr;

    // ---- per-lane addresses of the coalesced loads (unit 0; a later unit adds its block offset) --------------------------------------
    const int b0 = 8 * wave;                                               // first block of the wave's unit 0
    auto rowg = [&](int r) { const int x = rseg + r; return (int64_t) (x < Nseg ? x : Nseg - 1) * nb; };      // (the last tile of a matrix whose rows are not a multiple of 16: clamped, never stored)
    const int pc8 = lane & 7, pc16 = lane & 15, p2 = lane & 3;             // piece of a run: block (16-byte nibble runs), 16-byte piece (Q8_0), pair
    const char * aq_g[NAQ];
    #pragma unroll
    for (int k = 0; k < NAQ; k++) aq_g[k] = Q8 ? wbase + (rowg(4*k + (lane >> 4)) + b0) * 32 : wbase + (rowg(8*k + (lane >> 3)) + b0) * 16;
    const char * aqh_g = wbase + nbt * 16 + (rowg(lane >> 2) + b0) * 4;
    const char * ad_g  = wbase + nbt * (Q8 ? 32 : (Q5 ? 20 : 16)) + (rowg(lane >> 2) + b0) * 2;
    const size_t istride = dg_img_stride(WT, K);
    auto col_img = [&](int col, int & ti, int & Ti) {
        const int tc = col < T ? col : T - 1;                              // (columns >= T read column T - 1 again; never stored)
        const int gi = tc >> 3; ti = tc & 7; Ti = T - 8*gi < 8 ? T - 8*gi : 8;
        return (const char *) a.planes + (size_t) gi * istride;
    };
    const char * bq_g[4 * CG]; const char * bdx_g[CG];
    #pragma unroll
    for (int k = 0; k < 4 * CG; k++) {
        const int run = 8*k + (lane >> 3), col = run >> 1, plane = run & 1;
        int ti, Ti; const char * img = col_img(col, ti, Ti);
        bq_g[k] = img + (size_t) plane * Ti * nb * 16 + ((size_t) ti * nb + b0) * 16;
    }
    #pragma unroll
    for (int c = 0; c < CG; c++) {
        int ti, Ti; const char * img = col_img(16*c + (lane >> 2), ti, Ti);
        bdx_g[c] = img + (size_t) 2 * Ti * nb * 16 + ((size_t) ti * nb + b0) * 4;
    }
    // LDS addresses: stores (run order) and loads (operand order: lane = 16 kg + i, K-group kg = block (kg >> 1) of the pair, half kg & 1)
    char * aq_s[NAQ];
    #pragma unroll
    for (int k = 0; k < NAQ; k++) aq_s[k] = Q8 ? R + L::OFF_AQ + (4*k + (lane >> 4)) * L::AQ_ROW + pc16 * 16 : R + L::OFF_AQ + (8*k + (lane >> 3)) * L::AQ_ROW + pc8 * 16;
    char * aqh_s = R + L::OFF_AQH + (lane >> 2) * 40 + p2 * 8;
    char * ad_s  = R + L::OFF_AD + (lane >> 2) * 20 + p2 * 4;
    char * bq_s[4 * CG];
    #pragma unroll
    for (int k = 0; k < 4 * CG; k++) { const int run = 8*k + (lane >> 3); bq_s[k] = R + L::OFF_BQ + (run & 1) * L::BQ_PLANE + (run >> 1) * 144 + pc8 * 16; }
    char * bdx_s = R + L::OFF_BDX + (lane >> 2) * 40 + p2 * 8;
    const char * aq_l  = Q8 ? R + L::OFF_AQ + i * L::AQ_ROW + ((kg >> 1) * 2 + (kg & 1)) * 16 : R + L::OFF_AQ + i * L::AQ_ROW + (kg >> 1) * 16;
    const char * aqh_l = R + L::OFF_AQH + i * 40 + (kg >> 1) * 4;
    const char * ad_l  = R + L::OFF_AD + i * 20;
    const char * bq_l  = R + L::OFF_BQ + (kg & 1) * L::BQ_PLANE + i * 144 + (kg >> 1) * 16;
    const char * bdx_l = R + L::OFF_BDX + i * 40;

    // blocks of unit u that exist (wave-uniform): 8, or fewer at the end of a short row (K = 384: 12 blocks), or none
    auto unit_blocks = [&](int u) { const int left = nb - (b0 + LS*u); return left < 0 ? 0 : (left < 8 ? left : 8); };
    mx_unit<WT, CG> G;
    auto load_unit = [&](int u) {
        const int nv = unit_blocks(u);                                     // > 0 (caller)
        const int64_t ub = LS * u;
        // pieces beyond the row's end read its last block / pair again (never used)
        const int c8 = pc8 < nv ? pc8 : nv - 1, c16 = pc16 < 2*nv ? pc16 : 2*nv - 1, c2 = 2*p2 < nv ? p2 : (nv >> 1) - 1;
        #pragma unroll
        for (int k = 0; k < NAQ; k++) G.aq[k] = __builtin_nontemporal_load((const u32x4 *) (aq_g[k] + (Q8 ? ub * 32 + c16 * 16 : (ub + c8) * 16)));
        if constexpr (Q5) G.aqh = *(const uint2 *) (aqh_g + (ub + 2*c2) * 4); else G.aqh = make_uint2(0, 0);
        G.ad = *(const uint32_t *) (ad_g + (ub + 2*c2) * 2);
        #pragma unroll
        for (int k = 0; k < 4 * CG; k++) G.bq[k] = *(const u32x4 *) (bq_g[k] + (ub + c8) * 16);
        #pragma unroll
        for (int c = 0; c < CG; c++) G.bdx[c] = *(const uint2 *) (bdx_g[c] + (ub + 2*c2) * 4);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto store_unit = [&]() {
        #pragma unroll
        for (int k = 0; k < NAQ; k++) *(u32x4 *) aq_s[k] = G.aq[k];
        if constexpr (Q5) *(uint2 *) aqh_s = G.aqh;
        *(uint32_t *) ad_s = G.ad;
        #pragma unroll
        for (int k = 0; k < 4 * CG; k++) *(u32x4 *) bq_s[k] = G.bq[k];
        #pragma unroll
        for (int c = 0; c < CG; c++) *(uint2 *) (bdx_s + c * 16 * 40) = G.bdx[c];
        // the wave reads what its own lanes wrote: LDS operations of one wave complete in order; nothing may be moved across this point
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // ---- constants ------------------------------------------------------------------------------------------------------------------
    int mg; asm volatile("v_mov_b32 %0, 0x4b400000" : "=v"(mg));          // C operand of the integer MFMAs: D read as a float = 12582912 + sum
    const i32x4_t cmagic = { mg, mg, mg, mg };
    const f32x4_t zf = { 0.0f, 0.0f, 0.0f, 0.0f };
    const uint32_t m_lo = kg < 2 ? 0xFFFFFFFFu : 0u, m_hi = ~m_lo;        // block b lives in K-groups 0, 1 (lanes 0..31), block b + 1 in 2, 3
    const uint32_t m_s  = kg == 0 ? 0xFFFFFFFFu : 0u;                     // rank-1 scale MFMA: K-slot 0 = element 0 of lanes 0..15
    const int nsh = Q8 ? 0 : 4 * (kg & 1);                                 // this lane's nibble of every byte
    const int hsh = 16 * (kg & 1);                                         // ... and its 16 high bits (Q5_0)
    typedef int i32x2_t __attribute__((ext_vector_type(2)));

    load_unit(0);                                                          // (every wave of the launch has a unit 0)
    if constexpr (Q5) __syncthreads();                                     // the unpack table is complete
    store_unit();
    if (NU > 1 && unit_blocks(1) > 0) load_unit(1);

    // ---- units -> leaves: accA / accB = the chains of a pair's two leaves (k_gemv_q: one lane's acc over its units) ---------------------
    f32x4_t accA[4][CG], accB[4][CG];
    #pragma unroll
    for (int j = 0; j < 4; j++)
        #pragma unroll
        for (int c = 0; c < CG; c++) { accA[j][c] = zf; accB[j][c] = zf; }
    #pragma unroll
    for (int u = 0; u < NU; u++) {
        const int nv = unit_blocks(u);
        if (nv <= 0) break;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            if (2*j >= nv) continue;                                       // (wave-uniform) the row has no such blocks: the leaf's chain ends
            // A: this lane's 16 signed bytes
            const u32x4 q = *(const u32x4 *) (aq_l + j * (Q8 ? 64 : 32));
            uint32_t w[4] = { q[0], q[1], q[2], q[3] };
            if constexpr (Q5) {
                const uint32_t inv = (~*(const uint32_t *) (aqh_l + j * 8)) >> hsh;      // bit k set: element k of this half is negative (x - 16 = 0xF0 | nib)
                #pragma unroll
                for (int e = 0; e < 4; e++) w[e] = ((w[e] >> nsh) & 0x0F0F0F0Fu) | lut[(inv >> (4*e)) & 0xFu];
            } else if constexpr (WT == MI355X_TYPE_Q4_0) {
                #pragma unroll
                for (int e = 0; e < 4; e++) w[e] = ((((w[e] >> nsh) & 0x0F0F0F0Fu) | 0x80808080u) - 0x08080808u) ^ 0x80808080u;      // nib - 8 per byte (mmq.hip: q4_signed)
            }
            const i32x4_t a_lo = { (int) (w[0] & m_lo), (int) (w[1] & m_lo), (int) (w[2] & m_lo), (int) (w[3] & m_lo) };
            const i32x4_t a_hi = { (int) (w[0] & m_hi), (int) (w[1] & m_hi), (int) (w[2] & m_hi), (int) (w[3] & m_hi) };
            // scale operands: the pair's two f16 weight scales, K-slot 0 only
            const uint32_t dd = *(const uint32_t *) (ad_l + j * 4) & m_s;
            const i32x2_t sa0 = { (int) (dd & 0xFFFFu), 0 }, sa1 = { (int) (dd >> 16), 0 };
            #pragma unroll
            for (int c = 0; c < CG; c++) {
                const u32x4 bqv = *(const u32x4 *) (bq_l + c * 16 * 144 + j * 32);
                const float2 bdx = *(const float2 *) (bdx_l + c * 16 * 40 + j * 8);
                const i32x4_t bq = { (int) bqv[0], (int) bqv[1], (int) bqv[2], (int) bqv[3] };
                const i32x4_t S0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_lo, bq, cmagic, 0, 0, 0);
                const i32x4_t S1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hi, bq, cmagic, 0, 0, 0);
                const i32x2_t sb0 = { (int) (uint32_t) f2h(bdx.x), 0 }, sb1 = { (int) (uint32_t) f2h(bdx.y), 0 };      // exact: a Q8_0 scale is an f16 value
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
        if (u + 1 < NU && unit_blocks(u + 1) > 0) {
            store_unit();                                                  // (behind this unit's LDS reads: in order)
            if (u + 2 < NU && unit_blocks(u + 2) > 0) load_unit(u + 2);
        }
    }
    // butterfly steps lane ^ 1 (the pair), ^ 2, ^ 4 (the wave's pairs).  A leaf without blocks is +0 and x + 0 == x for every x these sums can
    // take (a sum of values that are not -0 is never -0), so adding the empty leaves as k_gemv_q does changes no bit.
    f32x4_t part[CG];
    #pragma unroll
    for (int c = 0; c < CG; c++)
        part[c] = ((accA[0][c] + accB[0][c]) + (accA[1][c] + accB[1][c])) + ((accA[2][c] + accB[2][c]) + (accA[3][c] + accB[3][c]));
    // ---- partial tiles -> LDS, at the head of the wave's own staging area: [column][MX_PSTRIDE], this lane's rows 4 kg .. 4 kg + 3 of column i + 16 c ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    #pragma unroll
    for (int c = 0; c < CG; c++) *(f32x4_t *) ((float *) R + (i + 16*c) * MX_PSTRIDE + 4*kg) = part[c];
    __syncthreads();

    // ---- combine over the waves (butterfly steps lane ^ 8, ^ 16, ^ 32), epilogue, stores ----------------------------------------------
    float * otile = (float *) (smem + 16 * 2 * MX_PSTRIDE * 4);           // [column][32] result tile of a planes-out workgroup: behind wave 0's partial tile
    auto finish = [&](const epi_t & e) {
        if (!e.ok) return;
        const int r = e.row & 15;                                          // (segments start at multiples of 16 rows)
        float P[8];
        #pragma unroll
        for (int w8 = 0; w8 < 8; w8++) P[w8] = w8 < nwt ? ((const float *) (smem + (size_t) (e.tl * nwt + w8) * L::SIZE))[e.t * MX_PSTRIDE + r] : 0.0f;
        float v = ((P[0] + P[1]) + (P[2] + P[3])) + ((P[4] + P[5]) + (P[6] + P[7]));
        if (sgr.bias)      v = v + e.bias;
        if (sgr.has_scale) v = v * sgr.scale;
        if (sgr.gelu)      v = gelu_lut(v, a.gelu_tab);
        if (has_res)       v = v + e.res;
        if (!(RTP == 2 && LS == 64 && a.planes_only)) {
            if (sgr.dst_f16) ((uint16_t *) e.dcol)[e.row] = f2h(v); else ((float *) e.dcol)[e.row] = v;
        }
        if constexpr (LS == 8) { if (e.mcol) e.mcol[e.row] = v; }
        if constexpr (RTP == 2 && LS == 64) otile[e.t*32 + e.tl*MX_ROWS + r] = v;
    };
    #pragma unroll
    for (int k = 0; k < EPI; k++) finish(ep[k]);
    for (int k = EPI; k * nthreads < NOUT; k++) finish(epi_prep(tid + k * nthreads));      // (narrow workgroups of small models only)
    if constexpr (RTP == 2 && LS == 64) {
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

template <int WT, int LS, int NU, int CG, int RTP, bool NSEG1>
static int mx_emit(mi355x_ctx * ctx, const MXArgs & k, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    static std::atomic<bool> attr_set[64];
    const int dev = ctx->device & 63;
    if (lds > 64 * 1024 && !attr_set[dev].load()) {
        // the ceiling, not this launch's size: `lds` grows with K (68.8 KB at K = 1280 ... 110 KB at 2048) and the attribute is set once per instantiation and device
        if (hipFuncSetAttribute((const void *) k_gemv_mx<WT, LS, NU, CG, RTP, NSEG1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) { (void) hipGetLastError(); return MI355X_E_UNSUPPORTED; }
        attr_set[dev].store(true);
    }
    return emit(ctx, LS == 8 ? "vocab_mx" : "gemv_mx", k_gemv_mx<WT, LS, NU, CG, RTP, NSEG1>, grid, block, lds, k, bytes, flops);
}

template <int WT, int CG>
static int mx_launch(mi355x_ctx * ctx, const MXArgs & k, bool vocab, int nu, bool pout, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    if (vocab) {
        switch (nu) {
            case 2: return mx_emit<WT, 8, 2, CG, MX_VOCAB_TILES, true>(ctx, k, grid, block, lds, bytes, flops);
            case 3: return mx_emit<WT, 8, 3, CG, MX_VOCAB_TILES, true>(ctx, k, grid, block, lds, bytes, flops);
            case 4: return mx_emit<WT, 8, 4, CG, MX_VOCAB_TILES, true>(ctx, k, grid, block, lds, bytes, flops);
            case 5: return mx_emit<WT, 8, 5, CG, MX_VOCAB_TILES, true>(ctx, k, grid, block, lds, bytes, flops);
        }
        return MI355X_E_UNSUPPORTED;
    }
    const bool nseg1 = k.nseg == 1;
    if (pout) {
        if (nu != 1 || !nseg1) return MI355X_E_UNSUPPORTED;
        return mx_emit<WT, 64, 1, CG, 2, true>(ctx, k, grid, block, lds, bytes, flops);
    }
    switch (nu) {
        case 1: return nseg1 ? mx_emit<WT, 64, 1, CG, 1, true>(ctx, k, grid, block, lds, bytes, flops) : mx_emit<WT, 64, 1, CG, 1, false>(ctx, k, grid, block, lds, bytes, flops);
        case 2: return nseg1 ? mx_emit<WT, 64, 2, CG, 1, true>(ctx, k, grid, block, lds, bytes, flops) : MI355X_E_UNSUPPORTED;
        case 3: return nseg1 ? mx_emit<WT, 64, 3, CG, 1, true>(ctx, k, grid, block, lds, bytes, flops) : MI355X_E_UNSUPPORTED;
    }
    return MI355X_E_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Q4_K (ggml-common.h:327-338; vec_dot q4_K x q8_K: ggml-cpu/quants.c:696-769).  k_gemv_q's unit for Q4_K is a 64-element chunk — 32 bytes whose low
// nibbles are sub-block 2c and whose high nibbles are sub-block 2c + 1 of a 256-element super-block — against the Q8_K activations of the chunk in
// four 16-byte planes.  That is exactly one MFMA pair: K-groups 0 / 1 = low nibbles of bytes 0..15 / 16..31 against planes 0 / 1, K-groups 2 / 3 = high
// nibbles against planes 2 / 3, issued twice with the other sub-block's half of A masked.  Then, per (row, column), k_gemv_q's own statements:
//     isum = sc_lo * S_lo + sc_hi * S_hi            msum = m_lo * bsum_lo + m_hi * bsum_hi          (integers: exact in f32, < 2^23)
//     acc  = fmaf(dx * d, isum, acc)                accm = fmaf(-dx * dmin, msum, accm)             leaf = acc + accm
// with the 6-bit sub-block scales / mins of the tile's 16 rows decoded once per unit into LDS, and the leaves summed in k_gemv_q's tree (leaf g =
// chunks g, g + 64).  A wave owns 8 neighbouring chunks (one unit = 512 features = two super-blocks) of its 16 rows.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <int CG> struct mx4k_lds {
    static constexpr int OFF_AQ  = 0;                                  // [16 rows][8 chunks x 32 B + 16]
    static constexpr int OFF_ASC = OFF_AQ + 16 * 272;                  // [16 rows][2 super-blocks x 12 B, padded to 32]: the packed 6-bit scales / mins
    static constexpr int OFF_ADM = OFF_ASC + 16 * 32;                  // [16 rows][2 x (f16 d, f16 dmin)]
    static constexpr int OFF_SC  = OFF_ADM + 16 * 8;                   // decoded: sc[16 sub-blocks][16 rows] f32 | m[16][16] | d[2][16] | dmin[2][16]
    static constexpr int OFF_M   = OFF_SC + 16 * 16 * 4;
    static constexpr int OFF_D   = OFF_M + 16 * 16 * 4;
    static constexpr int OFF_DMIN = OFF_D + 2 * 16 * 4;
    static constexpr int OFF_BQ  = OFF_DMIN + 2 * 16 * 4;              // [4 planes][16 CG columns][8 x 16 + 16]
    static constexpr int BQ_PLANE = 16 * CG * 144;
    static constexpr int OFF_BDX = OFF_BQ + 4 * BQ_PLANE;              // [16 CG columns][2 super-block scales] f32
    static constexpr int OFF_BBS = OFF_BDX + 16 * CG * 8;              // [16 CG columns][16 sub-block sums + 4] i32
    static constexpr int SIZE    = OFF_BBS + 16 * CG * 80;             // multiple of 16
};

// LS = 64: k_gemv_q's tree (lane g = chunks g, g + 64; the tile's waves own 8 neighbouring chunks each).  LS = 8: k_gemv8's tree at 8 lanes per row, the
// vocabulary projection (decode.hip: lane j = chunks j, j + 8, j + 16 chained, leaf = acc + accm, group_sum<8>): ONE wave per tile walks the whole row in
// units of 8 chunks, RTP = 4 tiles per workgroup.
template <int LS, int NU, int CG, int RTP, bool NSEG1>
__global__ void __launch_bounds__(512) k_gemv_mx4k(const MXArgs a) {
    typedef mx4k_lds<CG> L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nwt = a.nwt;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kg = lane >> 4;
    const int K = a.K, nch = K >> 6, nsb = K >> 8, T = a.T, ntot = a.ntot;
    const int tl = LS == 8 ? wave : 0, cw = LS == 8 ? 0 : wave;        // tile of the workgroup; place of the wave in the tile's chain
    constexpr int US = LS == 8 ? 8 : 64;                                   // chunks from one unit of a wave to its next
    const int row0 = (blockIdx.x * RTP + tl) * MX_ROWS;
    int s = 0;
    if constexpr (!NSEG1) {
        if (a.nseg > 1 && row0 >= a.row_start[1]) s = 1;
        if (a.nseg > 2 && row0 >= a.row_start[2]) s = 2;
    }
    const MXSeg & sgr = NSEG1 ? a.seg[0] : a.seg[s];
    const int seg0 = NSEG1 ? 0 : a.row_start[s];
    const int rseg = row0 - seg0;
    const char * wbase = (const char *) sgr.w;
    const int64_t nbt = sgr.nbt;                                          // super-blocks of the tensor
    const int Nseg = sgr.N;
    char * R = smem + (size_t) wave * L::SIZE;

    // ---- the epilogue's operands of this thread's first outputs, requested now ----
    typedef const MXArgs __attribute__((address_space(4))) * kargs_t;
    const kargs_t ka = (kargs_t) __builtin_amdgcn_kernarg_segment_ptr();
    const int nthreads = blockDim.x;
    constexpr int NOUT = RTP * MX_ROWS * 16 * CG;
    constexpr int EPI = 2;
    struct epi_t { int row, t, tl; float bias, res; void * dcol; float * mcol; bool ok; };
    auto epi_prep = [&](int o) {
        epi_t e;
        e.tl = RTP == 1 ? 0 : o / (MX_ROWS * 16 * CG);
        const int w = RTP == 1 ? o : o % (MX_ROWS * 16 * CG), r = w & 15;
        e.t = w >> 4;
        const int grow = (blockIdx.x * RTP + e.tl) * MX_ROWS + r;
        e.ok = o < NOUT && e.t < T && grow < ntot;
        e.row = e.ok ? grow - seg0 : 0;
        const int tc = e.ok ? e.t : 0;
        e.dcol = ka->cols.dst[NSEG1 ? 0 : s][tc];
        e.mcol = LS == 8 ? (float *) ka->cols.mirror[tc] : nullptr;
        const float * rcol = ka->cols.res[NSEG1 ? 0 : s][tc];
        e.bias = sgr.bias ? sgr.bias[e.row] : 0.0f;
        e.res  = rcol ? rcol[e.row] : 0.0f;
        return e;
    };
    epi_t ep[EPI];
    #pragma unroll
    for (int k = 0; k < EPI; k++) ep[k] = epi_prep(tid + k * nthreads);
    const bool has_res = ka->cols.res[NSEG1 ? 0 : s][0] != nullptr;

    // ---- per-lane addresses of the coalesced loads (unit 0) ----
    const int c0 = 8 * cw;                                                 // first chunk of the wave's unit 0
    auto rowi = [&](int r) { const int x = rseg + r; return x < Nseg ? x : Nseg - 1; };
    const int pc8 = lane & 7, pc16 = lane & 15, p4 = lane & 3;
    // weights: chunk ch of row r = 32 bytes at qs + (r * nsb + ch / 4) * 128 + (ch & 3) * 32: a unit's 8 chunks = 256 contiguous bytes of the row
    const char * aq_g[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) aq_g[k] = wbase + ((int64_t) rowi(4*k + (lane >> 4)) * nsb) * 128 + (int64_t) c0 * 32;
    const char * asc_g = wbase + nbt * 128 + ((int64_t) rowi(lane >> 2) * nsb + (c0 >> 2)) * 12;      // 2 super-blocks x 12 bytes = 6 dwords per row
    const char * adm_g = wbase + nbt * 140 + ((int64_t) rowi(lane >> 2) * nsb + (c0 >> 2)) * 4;
    const size_t istride = dg_img_stride(MI355X_TYPE_Q4_K, K);
    auto col_img = [&](int col, int & ti, int & Ti) {
        const int tc = col < T ? col : T - 1;
        const int gi = tc >> 3; ti = tc & 7; Ti = T - 8*gi < 8 ? T - 8*gi : 8;
        return (const char *) a.planes + (size_t) gi * istride;
    };
    // Q8_K image: q[4 planes][Ti][nch] uint4 | d[Ti][nsb] f32 | bsum[Ti][nsb * 8] i32
    const char * bq_g[8 * CG]; const char * bdx_g[CG]; const char * bbs_g[CG];
    #pragma unroll
    for (int k = 0; k < 8 * CG; k++) {
        const int run = 8*k + (lane >> 3), col = run >> 2, plane = run & 3;
        int ti, Ti; const char * img = col_img(col, ti, Ti);
        bq_g[k] = img + (((size_t) plane * Ti + ti) * nch + c0) * 16;
    }
    #pragma unroll
    for (int c = 0; c < CG; c++) {
        int ti, Ti; const char * img = col_img(16*c + (lane >> 2), ti, Ti);
        bdx_g[c] = img + (size_t) Ti * K + ((size_t) ti * nsb + (c0 >> 2)) * 4;
        bbs_g[c] = img + (size_t) Ti * K + (size_t) Ti * nsb * 4 + ((size_t) ti * nsb * 8 + (size_t) (c0 >> 2) * 8) * 4;
    }
    char * aq_s[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) aq_s[k] = R + L::OFF_AQ + (4*k + (lane >> 4)) * 272 + pc16 * 16;
    char * bq_s[8 * CG];
    #pragma unroll
    for (int k = 0; k < 8 * CG; k++) { const int run = 8*k + (lane >> 3); bq_s[k] = R + L::OFF_BQ + (run & 3) * L::BQ_PLANE + (run >> 2) * 144 + pc8 * 16; }
    const char * aq_l = R + L::OFF_AQ + i * 272 + (kg & 1) * 16;          // + chunk * 32: this lane's 16 bytes of the chunk (low nibbles kg < 2, high kg >= 2)
    const char * bq_l = R + L::OFF_BQ + kg * L::BQ_PLANE + i * 144;       // + column group * 16 * 144 + chunk * 16

    auto unit_chunks = [&](int u) { const int left = nch - (c0 + US*u); return left < 0 ? 0 : (left < 8 ? left : 8); };
    struct { u32x4 aq[4]; uint32_t asc[2]; uint32_t adm; u32x4 bq[8 * CG]; uint32_t bdx[CG]; u32x4 bbs[CG]; } G;
    auto load_unit = [&](int u) {
        const int nv = unit_chunks(u);                                     // > 0 (caller)
        const int64_t uc = US * u;                                         // chunk offset of the unit
        const int nsbu = (nv + 3) >> 2;                                    // super-blocks the unit touches (1 or 2)
        const int c8 = pc8 < nv ? pc8 : nv - 1, c16 = pc16 < 2*nv ? pc16 : 2*nv - 1;
        #pragma unroll
        for (int k = 0; k < 4; k++) G.aq[k] = __builtin_nontemporal_load((const u32x4 *) (aq_g[k] + uc * 32 + c16 * 16));
        // scales: dwords p4 and p4 + 4 (p4 < 2) of the row's 6; the second super-block may not exist
        { const int d0 = p4 < 3 * nsbu ? p4 : 0, d1 = 4 + (p4 & 1) < 3 * nsbu ? 4 + (p4 & 1) : 0;
          G.asc[0] = *(const uint32_t *) (asc_g + (uc >> 2) * 12 + d0 * 4); G.asc[1] = *(const uint32_t *) (asc_g + (uc >> 2) * 12 + d1 * 4); }
        G.adm = *(const uint32_t *) (adm_g + (uc >> 2) * 4 + ((p4 & 1) < nsbu ? (p4 & 1) : 0) * 4);
        #pragma unroll
        for (int k = 0; k < 8 * CG; k++) G.bq[k] = *(const u32x4 *) (bq_g[k] + (uc + c8) * 16);
        #pragma unroll
        for (int c = 0; c < CG; c++) {
            G.bdx[c] = *(const uint32_t *) (bdx_g[c] + (uc >> 2) * 4 + ((p4 & 1) < nsbu ? (p4 & 1) : 0) * 4);
            G.bbs[c] = *(const u32x4_a4 *) (bbs_g[c] + (uc >> 2) * 32 + (p4 < 2 * nsbu ? p4 : 0) * 16);    // 16 sums = 4 pieces of 4 (dword-aligned only: the sums start T * (K + 4 nsb) bytes into the image)
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto store_unit = [&]() {
        #pragma unroll
        for (int k = 0; k < 4; k++) *(u32x4 *) aq_s[k] = G.aq[k];
        *(uint32_t *) (R + L::OFF_ASC + (lane >> 2) * 32 + p4 * 4) = G.asc[0];
        if (p4 < 2) *(uint32_t *) (R + L::OFF_ASC + (lane >> 2) * 32 + (4 + p4) * 4) = G.asc[1];
        if (p4 < 2) *(uint32_t *) (R + L::OFF_ADM + (lane >> 2) * 8 + p4 * 4) = G.adm;
        #pragma unroll
        for (int k = 0; k < 8 * CG; k++) *(u32x4 *) bq_s[k] = G.bq[k];
        #pragma unroll
        for (int c = 0; c < CG; c++) {
            if (p4 < 2) *(uint32_t *) (R + L::OFF_BDX + (16*c + (lane >> 2)) * 8 + p4 * 4) = G.bdx[c];
            *(u32x4 *) (R + L::OFF_BBS + (16*c + (lane >> 2)) * 80 + p4 * 16) = G.bbs[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // decode the 6-bit scales / mins of the unit's 16 sub-blocks x 16 rows (get_scale_min_k4, ggml-quants.c:880-887): lane = (row, 4 sub-blocks)
        {
            const int row = lane >> 2, sbk = (lane >> 1) & 1, j0 = (lane & 1) * 4;          // super-block of the unit, first sub-block of this lane's four
            const uint32_t * rs = (const uint32_t *) (R + L::OFF_ASC + row * 32) + sbk * 3;
            const uint32_t s0 = rs[0], s1 = rs[1], s2 = rs[2];
            #pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                int sc, m; q4k_scale_min_w(j0 + jj, s0, s1, s2, sc, m);
                ((float *) (R + L::OFF_SC))[(sbk * 8 + j0 + jj) * 16 + row] = (float) sc;
                ((float *) (R + L::OFF_M))[(sbk * 8 + j0 + jj) * 16 + row]  = (float) m;
            }
            if ((lane & 1) == 0) {
                const uint32_t dm = ((const uint32_t *) (R + L::OFF_ADM + row * 8))[sbk];
                ((float *) (R + L::OFF_D))[sbk * 16 + row]    = h2f((uint16_t) (dm & 0xFFFF));
                ((float *) (R + L::OFF_DMIN))[sbk * 16 + row] = h2f((uint16_t) (dm >> 16));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    int mg; asm volatile("v_mov_b32 %0, 0x4b400000" : "=v"(mg));
    const i32x4_t cmagic = { mg, mg, mg, mg };
    const f32x4_t zf = { 0.0f, 0.0f, 0.0f, 0.0f };
    const uint32_t m_lo = kg < 2 ? 0xFFFFFFFFu : 0u, m_hi = ~m_lo;
    const int nsh = 4 * (kg >> 1);                                         // low nibbles for K-groups 0 / 1 (sub-block 2c), high for 2 / 3 (2c + 1)

    load_unit(0);
    store_unit();

    f32x4_t acc[8][CG], accm[8][CG];
    #pragma unroll
    for (int j = 0; j < 8; j++)
        #pragma unroll
        for (int c = 0; c < CG; c++) { acc[j][c] = zf; accm[j][c] = zf; }
    #pragma nounroll
    for (int u = 0; u < NU; u++) {
        const int nv = unit_chunks(u);
        if (nv <= 0) break;
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j >= nv) continue;                                         // (wave-uniform)
            const int sbk = j >> 2, jl = 2 * (j & 3);                      // super-block of the unit; sub-blocks jl (low nibbles), jl + 1 (high)
            const u32x4 q = *(const u32x4 *) (aq_l + j * 32);
            const i32x4_t wv = { (int) ((q[0] >> nsh) & 0x0F0F0F0Fu), (int) ((q[1] >> nsh) & 0x0F0F0F0Fu), (int) ((q[2] >> nsh) & 0x0F0F0F0Fu), (int) ((q[3] >> nsh) & 0x0F0F0F0Fu) };
            const i32x4_t a_lo = { (int) ((uint32_t) wv[0] & m_lo), (int) ((uint32_t) wv[1] & m_lo), (int) ((uint32_t) wv[2] & m_lo), (int) ((uint32_t) wv[3] & m_lo) };
            const i32x4_t a_hi = { (int) ((uint32_t) wv[0] & m_hi), (int) ((uint32_t) wv[1] & m_hi), (int) ((uint32_t) wv[2] & m_hi), (int) ((uint32_t) wv[3] & m_hi) };
            // this lane's rows 4 kg .. 4 kg + 3: sub-block scales / mins, super-block d / dmin
            const f32x4_t sc_lo = *(const f32x4_t *) (R + L::OFF_SC + ((sbk * 8 + jl) * 16 + 4*kg) * 4), sc_hi = *(const f32x4_t *) (R + L::OFF_SC + ((sbk * 8 + jl + 1) * 16 + 4*kg) * 4);
            const f32x4_t mn_lo = *(const f32x4_t *) (R + L::OFF_M  + ((sbk * 8 + jl) * 16 + 4*kg) * 4), mn_hi = *(const f32x4_t *) (R + L::OFF_M  + ((sbk * 8 + jl + 1) * 16 + 4*kg) * 4);
            const f32x4_t dwv = *(const f32x4_t *) (R + L::OFF_D + (sbk * 16 + 4*kg) * 4), dmv = *(const f32x4_t *) (R + L::OFF_DMIN + (sbk * 16 + 4*kg) * 4);
            #pragma unroll
            for (int c = 0; c < CG; c++) {
                const u32x4 bqv = *(const u32x4 *) (bq_l + c * 16 * 144 + j * 16);
                const i32x4_t bq = { (int) bqv[0], (int) bqv[1], (int) bqv[2], (int) bqv[3] };
                const i32x4_t S0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_lo, bq, cmagic, 0, 0, 0);
                const i32x4_t S1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hi, bq, cmagic, 0, 0, 0);
                const float dxv = *(const float *) (R + L::OFF_BDX + (16*c + i) * 8 + sbk * 4);
                const int * bsp = (const int *) (R + L::OFF_BBS + (16*c + i) * 80) + sbk * 8 + jl;
                const float bs_lo = (float) bsp[0], bs_hi = (float) bsp[1];
                const f32x4_t F0 = __builtin_bit_cast(f32x4_t, S0), F1 = __builtin_bit_cast(f32x4_t, S1);
                #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float isum = fmaf(sc_lo[r], F0[r] - 12582912.0f, sc_hi[r] * (F1[r] - 12582912.0f));      // exact: integers below 2^23
                    const float msum = fmaf(mn_lo[r], bs_lo, mn_hi[r] * bs_hi);
                    acc[j][c][r]  = fmaf(dxv * dwv[r], isum, acc[j][c][r]);            // k_gemv_q: acc = fmaf(dxv*dw, (float) isum, acc)
                    accm[j][c][r] = fmaf(-dxv * dmv[r], msum, accm[j][c][r]);          //           accm = fmaf(-dxv*dminw, (float) msum, accm)
                }
            }
        }
        if (u + 1 < NU && unit_chunks(u + 1) > 0) {                        // (the second unit exists only at K > 4096: fetched after the first is consumed — 128 accumulators leave no room to hold it in flight)
            load_unit(u + 1);
            store_unit();
        }
    }
    // leaf = acc + accm, then k_gemv_q's butterfly over the wave's 8 leaves (lane ^ 1, ^ 2, ^ 4)
    f32x4_t part[CG];
    #pragma unroll
    for (int c = 0; c < CG; c++) {
        f32x4_t lf[8];
        #pragma unroll
        for (int j = 0; j < 8; j++) lf[j] = acc[j][c] + accm[j][c];
        part[c] = ((lf[0] + lf[1]) + (lf[2] + lf[3])) + ((lf[4] + lf[5]) + (lf[6] + lf[7]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    #pragma unroll
    for (int c = 0; c < CG; c++) *(f32x4_t *) ((float *) R + (i + 16*c) * MX_PSTRIDE + 4*kg) = part[c];
    __syncthreads();

    auto finish = [&](const epi_t & e) {
        if (!e.ok) return;
        const int r = e.row & 15;
        float P[8];
        #pragma unroll
        for (int w8 = 0; w8 < 8; w8++) P[w8] = w8 < nwt ? ((const float *) (smem + (size_t) (e.tl * nwt + w8) * L::SIZE))[e.t * MX_PSTRIDE + r] : 0.0f;
        float v = LS == 8 ? P[0] : ((P[0] + P[1]) + (P[2] + P[3])) + ((P[4] + P[5]) + (P[6] + P[7]));
        if (sgr.bias)      v = v + e.bias;
        if (sgr.has_scale) v = v * sgr.scale;
        if (sgr.gelu)      v = gelu_lut(v, a.gelu_tab);
        if (has_res)       v = v + e.res;
        if (sgr.dst_f16) ((uint16_t *) e.dcol)[e.row] = f2h(v); else ((float *) e.dcol)[e.row] = v;
        if constexpr (LS == 8) { if (e.mcol) e.mcol[e.row] = v; }
    };
    #pragma unroll
    for (int k = 0; k < EPI; k++) finish(ep[k]);
    for (int k = EPI; k * nthreads < NOUT; k++) finish(epi_prep(tid + k * nthreads));
}

template <int LS, int NU, int CG, int RTP, bool NSEG1>
static int mx4k_emit(mi355x_ctx * ctx, const MXArgs & k, dim3 grid, dim3 block, uint32_t lds, double bytes, double flops) {
    static std::atomic<bool> attr_set[64];
    const int dev = ctx->device & 63;
    if (lds > 64 * 1024 && !attr_set[dev].load()) {
        if (hipFuncSetAttribute((const void *) k_gemv_mx4k<LS, NU, CG, RTP, NSEG1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) { (void) hipGetLastError(); return MI355X_E_UNSUPPORTED; }
        attr_set[dev].store(true);
    }
    return emit(ctx, LS == 8 ? "vocab_mx" : "gemv_mx", k_gemv_mx4k<LS, NU, CG, RTP, NSEG1>, grid, block, lds, k, bytes, flops);
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
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0 && wt != MI355X_TYPE_Q4_K) return MI355X_E_UNSUPPORTED;
    const bool q4k = wt == MI355X_TYPE_Q4_K;
    if (K <= 0 || K % 64 || (q4k && K % 256)) return MI355X_E_UNSUPPORTED;   // pairs of blocks / whole super-blocks
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
        o.w = g.w; o.nbt = q4k ? (int64_t) g.N * (K / 256) : (int64_t) g.N * nb; o.bias = g.ep.bias; o.scale = g.ep.scale; o.has_scale = g.ep.has_scale; o.gelu = g.ep.gelu; o.dst_f16 = g.dst_type == MI355X_TYPE_F16; o.N = g.N;
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
    if (q4k && d->planes_out) return MI355X_E_UNSUPPORTED;                 // (as k_gemv_q: Q4_K has no planes-out form)
    // Q4_K's vocabulary projection runs k_gemv8 image by image; its tree is the 8-lanes-per-row one exactly when the launcher picks 8 lanes (decode.hip: mi355x_gemv)
    if (q4k && vocab && (int64_t) ntot * 8 / 64 < (int64_t) ctx->n_cu * 8) return MI355X_E_UNSUPPORTED;
    bool mirror = false;
    int nu, nwt, rtp = 1;
    const bool pout = d->planes_out != nullptr;
    if (vocab) {
        // k_vocab's shapes and tree (decode_q.hip: mi355x_vocab): 8 lanes per row, lane j = blocks j, j + 8, ...
        const mi355x_gemv_seg & g = d->seg[0];
        if (d->nseg != 1 || pout || K % 256 || K > 2048 || K / 256 < 2 || K / 256 > 5) return MI355X_E_UNSUPPORTED;
        if (q4k && K > 1536) return MI355X_E_UNSUPPORTED;                  // three units of 8 chunks
        if (g.ep.bias || g.ep.has_scale || g.ep.gelu || g.ep.residual || g.dst_type != MI355X_TYPE_F32) return MI355X_E_UNSUPPORTED;
        mirror = d->cols && d->cols->mirror[0];
        for (int t = 0; t < T; t++) { k.cols.mirror[t] = mirror ? d->cols->mirror[t] : nullptr; if (mirror && !k.cols.mirror[t]) return MI355X_E_UNSUPPORTED; }
        nu = K / 256; nwt = 1; rtp = MX_VOCAB_TILES;
    } else {
        const int units = q4k ? K / 64 : nb;                               // k_gemv_q's lane units: 64-element chunks (Q4_K) or 32-element blocks
        nu = (units + 63) / 64;
        if (nu > (q4k ? 2 : 3)) return MI355X_E_UNSUPPORTED;
        nwt = ((units < 64 ? units : 64) + 7) / 8;
        if (pout) {
            if (d->nseg != 1 || ntot % 32 || nu != 1 || ((uintptr_t) d->planes_out % 16)) return MI355X_E_UNSUPPORTED;
            k.planes_out = d->planes_out; k.planes_only = d->planes_out_only; rtp = 2;
        }
    }
    k.nwt = nwt;
    // Q4_K at K > 4096 (two units, 8 waves): two column groups do not fit the LDS — columns 16 .. T - 1 (images 2, 3) go in a second launch
    const bool q4k_halves = q4k && (vocab || nu == 2) && T > 16;           // (the vocabulary form holds 128 chained accumulators across its units' loads: one column group)
    const int cg = T > 16 && !q4k_halves ? 2 : 1;
    const int ntiles = (ntot + MX_ROWS - 1) / MX_ROWS;              // (a single segment may end inside its last tile: the vocabulary's 51864 / 51865 / 51866 rows)
    const dim3 grid((ntiles + rtp - 1) / rtp), block(64 * nwt * rtp);
    if (pout && (int) block.x < 16 * cg * 8) return MI355X_E_UNSUPPORTED;   // the planes-out pass needs 8 threads per column
    const size_t per_wave = q4k ? (cg == 2 ? mx4k_lds<2>::SIZE : mx4k_lds<1>::SIZE)
                          : wt == MI355X_TYPE_Q8_0 ? (cg == 2 ? mx_lds<MI355X_TYPE_Q8_0, 2>::SIZE : mx_lds<MI355X_TYPE_Q8_0, 1>::SIZE)
                                                   : (cg == 2 ? mx_lds<MI355X_TYPE_Q5_0, 2>::SIZE : mx_lds<MI355X_TYPE_Q5_0, 1>::SIZE);
    const uint32_t lds = (uint32_t) ((size_t) rtp * nwt * per_wave);
    if (lds > 160 * 1024 - 256) return MI355X_E_UNSUPPORTED;
    const double bytes = wbytes + (double) dg_planes_bytes(wt, K, T) + (double) ntot*T*4;
    const double flops = 2.0 * ntot * K * T;
    int rc;
    if (q4k) {
        if (d->nseg != 1 && (vocab || nu == 2)) return MI355X_E_UNSUPPORTED;
        if (!vocab && ntot % MX_ROWS) return MI355X_E_UNSUPPORTED;
        auto go = [&](double share) {
            const double by = bytes * share, fl = flops * share;
            if (vocab)   return mx4k_emit<8, 3, 1, MX_VOCAB_TILES, true>(ctx, k, grid, block, lds, by, fl);
            if (nu == 2) return mx4k_emit<64, 2, 1, 1, true>(ctx, k, grid, block, lds, by, fl);
            const bool n1 = d->nseg == 1;
            return cg == 2 ? (n1 ? mx4k_emit<64, 1, 2, 1, true>(ctx, k, grid, block, lds, by, fl) : mx4k_emit<64, 1, 2, 1, false>(ctx, k, grid, block, lds, by, fl))
                           : (n1 ? mx4k_emit<64, 1, 1, 1, true>(ctx, k, grid, block, lds, by, fl) : mx4k_emit<64, 1, 1, 1, false>(ctx, k, grid, block, lds, by, fl));
        };
        if (!q4k_halves) rc = go(1.0);
        else {
            k.T = 16;
            rc = go(16.0 / T);
            if (rc) return rc;
            k.T = T - 16;
            k.planes = (const char *) d->x_planes + 2 * dg_img_stride(MI355X_TYPE_Q4_K, K);
            for (int t = 0; t < T - 16; t++) { k.cols.dst[0][t] = k.cols.dst[0][t + 16]; k.cols.res[0][t] = k.cols.res[0][t + 16]; k.cols.mirror[t] = k.cols.mirror[t + 16]; }
            rc = go((double) (T - 16) / T);
        }
        if (rc == 0 && mirror) ctx->last_mirrored = 1;
        return rc;
    }
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
