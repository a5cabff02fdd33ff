// One launch per decoder layer for  LayerNorm -> Q / K / V projections -> KV-cache store -> self-attention  of a single-token step
// (src/whisper.cpp:2550-2660: norm, mul_mat x 3 with bias / scale, ggml_cpy into the caches, ggml_flash_attn_ext), T = 1.
//
// ONE WORKGROUP PER HEAD.  Head h needs rows [64h, 64h + 64) of W_q, W_k and W_v (192 rows, 170 KB of Q5_0 at n_state = 1280), the
// head's slice of the K / V caches (n_kv x 2 x 128 bytes) and nothing any other head produces: no hand-off between workgroups, so the
// dependent launch between the projections and the attention (one of eight per layer, ~4 us each on this stack) disappears instead
// of being replaced by an in-kernel barrier across the chip.  What it costs is that 20 CUs pull what 240 workgroups pulled before:
// measured +0.7 us for 176 KB per workgroup against 15 KB (scripts/ingest_probe.hip, profiles/r03b_ingest_probe.txt).
//
//   phase 0  every load of the kernel is requested: the head's cache rows by LDS-DMA (no registers: they land in LDS while the
//            projections run), x / LayerNorm vectors / mask values / bias values, then the 192 weight rows (RW rows per wave)
//   phase 1  LayerNorm + Q8_0 quantization of x into LDS planes      — statement for statement k_gemv_row's MODE 1 (decode.hip)
//   phase 2  RW integer-dot rows per wave, bias / scale epilogue; q -> LDS (f32), new k / v rows -> cache (f16) and LDS
//   phase 3  attention over the LDS copy of the cache, 32 keys per wave, with the new row taken from LDS;  per-wave records merged
//            four waves at a time into the 128-key partial records of k_fattn_dec                        — same statements as there
// Every arithmetic statement equals the two-kernel path's, so the partial records (and the cache rows) are bit-identical to
// mi355x_gemv_fused + mi355x_flash_attn_partial (tests/test_gpu_head.py); the output projection consumes the records unchanged.
#include "decode_common.h"
#include <atomic>

struct SHArgs {
    const float * x; int K; float eps; const float * ln_w; const float * ln_b;
    DGSeg seg[3];                        // 0 = q, 1 = k, 2 = v; dst of 1 / 2 = this step's rows in the caches (f16), dst of 0 = f32 or NULL
    const char * kc; const char * vc;    // caches as flash_attn_ext sees them: [64, n_kv, H] f16
    int64_t k_nb1, k_nb2, v_nb1, v_nb2;
    const char * mask;                   // f16 row of the query or NULL
    float fa_scale; int n_kv, new_key, nparts;
    float * part_o; float * part_ml;
    const uint16_t * valid;              // any valid device address (stands in for absent vectors)
};

#define SH_HDR 1024                      // red[2][8] floats | q[64] f32 | knew[64] f16 | vnew[64] f16

// 1 KiB per wave instruction: lane l's 16 bytes land at LDS byte address lds_dst + 16 l.  The statement owns M0.
__device__ __forceinline__ void sh_glds16(const char * gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int WT, int NW, int ND>           // ND = LDS-DMA pairs per wave: covers n_kv <= 8 * NW * ND keys
__global__ void __launch_bounds__(NW*64) k_self_head(const SHArgs a) {
    constexpr int RW = 192 / NW;                                         // weight rows per wave
    // Q8_0 rows are 9 registers per lane each and three loads: all 192 rows of a head at once fit neither the 128 registers of a
    // 16-wave workgroup nor (24 rows x 3 loads per lane) the 6-bit vmcnt of an 8-wave one.  Q8_0 runs with 8 waves and requests the
    // second half of a wave's rows when phase 1 is over (that latency is paid once, ~1 us)
    constexpr int RB = WT == MI355X_TYPE_Q8_0 ? RW / 2 : RW;             // rows requested in phase 0
    constexpr int NPASS = 16 / NW;                                       // attention passes: 16 waves' worth of keys (512) at most
    // two LDS objects: the LDS-DMA target (the head's cache rows) and everything the phases read and write themselves.  With one
    // object hipcc orders the first ds_write of phase 1 behind `s_waitcnt vmcnt(0)` — an LDS-DMA is a pending LDS write it cannot tell
    // apart from the planes — which also waits for every weight row requested after it.
    __shared__ __attribute__((aligned(1024))) char kvl[512*256];
    __shared__ __attribute__((aligned(1024))) char smem[SH_HDR + 3072 + 16*64*4 + 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x;
    const int K = a.K, nb = K >> 5, K4 = K >> 2;
    const int n_kv = a.n_kv;
    // waves that hold data in the mat-vec kernel (same order of the LayerNorm partial sums); computed up here: a basic-block boundary
    // between the load burst and the first use of x costs an s_waitcnt vmcnt(0), i.e. the LayerNorm would wait for the weights
    const int nwd_ = (K4 + 63) >> 6;
    const int nwd = nwd_ < 4 ? 4 : nwd_;
    float * red = (float *) smem;
    float * qv = (float *) (smem + 128);
    uint16_t * knew = (uint16_t *) (smem + 384), * vnew = (uint16_t *) (smem + 512);
    uint32_t * lo = (uint32_t *) (smem + SH_HDR);
    uint32_t * hi = lo + (size_t) nb*4;
    float * dx = (float *) (hi + (size_t) nb*4);
    int * sx = (int *) (dx + nb);
    constexpr size_t planes = 3072;                                      // K <= 2048: 64 blocks x 40 bytes
    float * wo = (float *) (smem + SH_HDR + planes);                     // [16][64]
    float * wml = wo + 16*64;                                            // [16][2]
    char * Kl = kvl;
    char * Vl = Kl + (size_t) n_kv * 128;

    // ---- phase 0: the load burst -------------------------------------------------------------------------------------------------
    const int e4c = tid < K4 ? tid : K4 - 1;
    const float4 xr = *(const float4 *) ((const char *) a.x + (size_t) e4c*16);
    __builtin_amdgcn_sched_barrier(0);
    const float4 lw = *(const float4 *) (a.ln_w + e4c*4), lb = *(const float4 *) (a.ln_b + e4c*4);
    // mask values of this lane's keys in each attention pass
    const int kg = lane >> 3, dc = lane & 7;
    uint16_t mkh[NPASS][4];
    {
        const char * mbase = a.mask ? a.mask : (const char *) a.valid;
        #pragma unroll
        for (int ps = 0; ps < NPASS; ps++)
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = (ps*NW + wave)*32 + kg + 8*i, kc = key < n_kv ? key : n_kv - 1;
                mkh[ps][i] = *(const uint16_t *) (mbase + (a.mask ? (int64_t) kc*2 : 0));
            }
    }
    // lane r < RW finishes row r of this wave: its segment, row in the head, bias
    const int rl = wave*RW + (lane < RW ? lane : RW - 1);
    const int sl = rl >> 6, rr = rl & 63;
    const float * sbias = sl == 0 ? a.seg[0].bias : (sl == 1 ? a.seg[1].bias : a.seg[2].bias);
    const float bias_v = *(sbias ? sbias + h*64 + rr : (const float *) a.valid);
    __builtin_amdgcn_sched_barrier(0);
    wblk<WT> wr[RW];
    auto request_rows = [&](int i0, int i1) {
        const int gc = lane < nb ? lane : nb - 1;
        #pragma unroll
        for (int i = i0; i < i1; i++) {
            const int R = wave*RW + i, s = R >> 6;                       // wave-uniform
            const DGSeg & sg = a.seg[s];
            wblk_load<WT>(wr[i], (const char *) sg.w, sg.nbt, (int64_t) (h*64 + (R & 63)) * nb + gc);
        }
    };
    request_rows(0, RB);
    __builtin_amdgcn_sched_barrier(0);
    // cache rows of this head -> LDS by LDS-DMA: one instruction = 8 keys x 128 bytes (lane -> key 8j + lane/8, 16-byte chunk lane%8).
    // Issued LAST and through inline asm: hipcc cannot count vmcnt once LDS-DMA and ordinary loads are in flight together (it waits
    // vmcnt(0) at the first use of x, i.e. the LayerNorm would start after the last weight row has arrived).  Hidden from its
    // bookkeeping they only make its counted waits a little conservative; the wait that matters for them is the explicit vmcnt(0)
    // in front of the barrier before phase 3.  ND pairs per wave, unconditional; groups past the end are clamped to the last one
    // (the same bytes land on the same LDS addresses again).
    {
        const int ndma = n_kv >> 3;
        const char * gk = a.kc + (int64_t) h*a.k_nb2 + (int64_t) (lane >> 3)*a.k_nb1 + (lane & 7)*16;
        const char * gv = a.vc + (int64_t) h*a.v_nb2 + (int64_t) (lane >> 3)*a.v_nb1 + (lane & 7)*16;
        const unsigned kl0 = (unsigned) (size_t) Kl, vl0 = (unsigned) (size_t) Vl;
        #pragma unroll
        for (int u = 0; u < ND; u++) {
            const int j0 = wave + u*NW, j = j0 < ndma ? j0 : ndma - 1;
            sh_glds16(gk + (int64_t) j*8*a.k_nb1, __builtin_amdgcn_readfirstlane(kl0 + j*1024));
            sh_glds16(gv + (int64_t) j*8*a.v_nb1, __builtin_amdgcn_readfirstlane(vl0 + j*1024));
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 1: ggml_norm (ggml-cpu/ops.cpp:3698-3765) + affine, two passes over the registers; Q8_0 planes into LDS ------------------
    {
        float p = 0.0f;
        if (tid < K4) p += (xr.x + xr.y) + (xr.z + xr.w);
        p = wave_sum(p);
        if (lane == 0 && wave < 8) red[wave] = p;
        __syncthreads();
        float pw[8];
        #pragma unroll
        for (int w = 0; w < 8; w++) pw[w] = red[w];
        float rs = 0.0f;
        #pragma unroll
        for (int w = 0; w < 8; w++) rs += w < nwd ? pw[w] : 0.0f;
        const float mean = rs / K;
        p = 0.0f;
        if (tid < K4) {
            const float d0 = xr.x - mean, d1 = xr.y - mean, d2 = xr.z - mean, d3 = xr.w - mean;
            p += (d0*d0 + d1*d1) + (d2*d2 + d3*d3);
        }
        p = wave_sum(p);
        if (lane == 0 && wave < 8) red[8 + wave] = p;
        __syncthreads();
        #pragma unroll
        for (int w = 0; w < 8; w++) pw[w] = red[8 + w];
        rs = 0.0f;
        #pragma unroll
        for (int w = 0; w < 8; w++) rs += w < nwd ? pw[w] : 0.0f;
        const float rstd = 1.0f / sqrtf(rs / K + a.eps);
        if (tid < K4) {
            float o[4] = { (xr.x - mean) * rstd, (xr.y - mean) * rstd, (xr.z - mean) * rstd, (xr.w - mean) * rstd };
            o[0] = o[0]*lw.x; o[1] = o[1]*lw.y; o[2] = o[2]*lw.z; o[3] = o[3]*lw.w;
            o[0] = o[0]+lb.x; o[1] = o[1]+lb.y; o[2] = o[2]+lb.z; o[3] = o[3]+lb.w;
            dg_q8_0_store(o, tid*4, 0, nb, lo, hi, dx, sx);
        }
        // empty records for the attention waves past the keys (k_fattn_dec leaves m = -1e30, l = 0, o = 0 there)
        for (int idx = tid; idx < 16*64; idx += NW*64) if ((idx >> 6)*32 >= n_kv) wo[idx] = 0.0f;
        if (tid < 16 && tid*32 >= n_kv) { wml[tid*2] = -1e30f; wml[tid*2 + 1] = 0.0f; }
    }
    __syncthreads();

    // ---- phase 2: RW rows per wave; lane handles block `lane` of every row (clamped duplicates carry weight 0) ------------------------------
    if constexpr (RB < RW) { request_rows(RB, RW); __builtin_amdgcn_sched_barrier(0); }
    {
        const int gc = lane < nb ? lane : nb - 1;
        const uint4 al = ((const uint4 *) lo)[gc], ah = ((const uint4 *) hi)[gc];
        constexpr int off = WT == MI355X_TYPE_Q5_0 ? 16 : (WT == MI355X_TYPE_Q4_0 ? 8 : 0);
        const int sxv = off ? sx[gc] : 0;
        const float dxv = dx[gc];
        float acc[RW];
        #pragma unroll
        for (int i = 0; i < RW; i++) {
            uint32_t vlo[4], vhi[4];
            wblk_unpack<WT>(wr[i], vlo, vhi);
            const float dw = lane < nb ? h2f(wr[i].d) : 0.0f;
            int sum = 0;
            sum = __builtin_amdgcn_sdot4((int) vlo[0], (int) al.x, sum, false);
            sum = __builtin_amdgcn_sdot4((int) vlo[1], (int) al.y, sum, false);
            sum = __builtin_amdgcn_sdot4((int) vlo[2], (int) al.z, sum, false);
            sum = __builtin_amdgcn_sdot4((int) vlo[3], (int) al.w, sum, false);
            sum = __builtin_amdgcn_sdot4((int) vhi[0], (int) ah.x, sum, false);
            sum = __builtin_amdgcn_sdot4((int) vhi[1], (int) ah.y, sum, false);
            sum = __builtin_amdgcn_sdot4((int) vhi[2], (int) ah.z, sum, false);
            sum = __builtin_amdgcn_sdot4((int) vhi[3], (int) ah.w, sum, false);
            if (off) sum -= off * sxv;
            acc[i] = fmaf(dw * dxv, (float) sum, 0.0f);
        }
        #pragma unroll
        for (int i = 0; i < RW; i++) acc[i] = wave_sum(acc[i]);
        float v = acc[0];
        #pragma unroll
        for (int i = 1; i < RW; i++) v = (lane == i) ? acc[i] : v;
        if (lane < RW) {
            const bool hs = sl == 0 ? a.seg[0].has_scale != 0 : (sl == 1 ? a.seg[1].has_scale != 0 : a.seg[2].has_scale != 0);
            const float sc = sl == 0 ? a.seg[0].scale : (sl == 1 ? a.seg[1].scale : a.seg[2].scale);
            if (sbias) v = v + bias_v;
            if (hs)    v = v * sc;
            if (sl == 0) {
                qv[rr] = v;
                if (a.seg[0].dst) ((float *) a.seg[0].dst)[h*64 + rr] = v;
            } else {
                const uint16_t hv = f2h(v);
                (sl == 1 ? knew : vnew)[rr] = hv;
                ((uint16_t *) (sl == 1 ? a.seg[1].dst : a.seg[2].dst))[h*64 + rr] = hv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this wave's LDS-DMA has landed
    __syncthreads();                                                    // ... and everyone's; q / new rows visible

    // ---- phase 3: attention, 32 keys per wave (k_fattn_dec<1>, decode.hip) --------------------------------------------------------------------
    {
        float qf[8];
        {
            const float4 q0 = *(const float4 *) (qv + dc*8), q1 = *(const float4 *) (qv + dc*8 + 4);
            qf[0] = round_f16(q0.x); qf[1] = round_f16(q0.y); qf[2] = round_f16(q0.z); qf[3] = round_f16(q0.w);
            qf[4] = round_f16(q1.x); qf[5] = round_f16(q1.y); qf[6] = round_f16(q1.z); qf[7] = round_f16(q1.w);
        }
        const uint4 kn = *(const uint4 *) ((const char *) knew + dc*16), vn = *(const uint4 *) ((const char *) vnew + dc*16);
        #pragma unroll
        for (int ps = 0; ps < NPASS; ps++) {
            const int aw = ps*NW + wave;
            if (aw*32 >= n_kv) continue;                                 // wave-uniform
            const int kbeg = aw*32;
            uint4 kr[4], vr[4];
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const int key = kbeg + kg + 8*i;
                const uint4 kl = *(const uint4 *) (Kl + (size_t) key*128 + dc*16), vl = *(const uint4 *) (Vl + (size_t) key*128 + dc*16);
                const bool isnew = key == a.new_key;
                kr[i] = make_uint4(isnew ? kn.x : kl.x, isnew ? kn.y : kl.y, isnew ? kn.z : kl.z, isnew ? kn.w : kl.w);
                vr[i] = make_uint4(isnew ? vn.x : vl.x, isnew ? vn.y : vl.y, isnew ? vn.z : vl.z, isnew ? vn.w : vl.w);
            }
            float sc[4];
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t w[4] = { kr[i].x, kr[i].y, kr[i].z, kr[i].w };
                float kf[8];
                #pragma unroll
                for (int e = 0; e < 4; e++) { kf[2*e] = h2f((uint16_t) (w[e] & 0xFFFF)); kf[2*e+1] = h2f((uint16_t) (w[e] >> 16)); }
                float s = 0.0f;
                #pragma unroll
                for (int e = 0; e < 8; e++) s = fmaf(kf[e], qf[e], s);
                s = group_sum<8>(s);
                sc[i] = s * a.fa_scale + (a.mask ? h2f(mkh[ps][i]) : 0.0f);     // (keys < n_kv: n_kv is a multiple of 32)
            }
            float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            m = stride8_max(m);
            m = fmaxf(m, -1e30f);
            float l = 0.0f, o[8];
            #pragma unroll
            for (int e = 0; e < 8; e++) o[e] = 0.0f;
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const float pk = __expf(sc[i] - m);
                l += pk;
                const uint32_t w[4] = { vr[i].x, vr[i].y, vr[i].z, vr[i].w };
                #pragma unroll
                for (int e = 0; e < 4; e++) {
                    o[2*e]   = fmaf(pk, h2f((uint16_t) (w[e] & 0xFFFF)), o[2*e]);
                    o[2*e+1] = fmaf(pk, h2f((uint16_t) (w[e] >> 16)),    o[2*e+1]);
                }
            }
            l = stride8_sum(l);
            #pragma unroll
            for (int e = 0; e < 8; e++) o[e] = stride8_sum(o[e]);
            if (kg == 0) {
                *(float4 *) (wo + aw*64 + dc*8)     = make_float4(o[0], o[1], o[2], o[3]);
                *(float4 *) (wo + aw*64 + dc*8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
                if (dc == 0) { wml[aw*2] = m; wml[aw*2 + 1] = l; }
            }
        }
    }
    __syncthreads();
    // merge four waves per 128-key record (k_fattn_dec's merge)
    for (int idx = tid; idx < a.nparts*64; idx += NW*64) {
        const int p = idx >> 6, d = idx & 63;
        const float m0 = wml[(4*p)*2], m1 = wml[(4*p + 1)*2], m2 = wml[(4*p + 2)*2], m3 = wml[(4*p + 3)*2];
        const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float w0 = __expf(m0 - M), w1 = __expf(m1 - M), w2 = __expf(m2 - M), w3 = __expf(m3 - M);
        const float o = fmaf(w3, wo[(4*p + 3)*64 + d], fmaf(w2, wo[(4*p + 2)*64 + d], fmaf(w1, wo[(4*p + 1)*64 + d], w0 * wo[(4*p)*64 + d])));
        const int64_t rec = (int64_t) h * a.nparts + p;
        a.part_o[rec*64 + d] = o;
        if (d == 0) {
            a.part_ml[rec*2]     = M;
            a.part_ml[rec*2 + 1] = fmaf(w3, wml[(4*p + 3)*2 + 1], fmaf(w2, wml[(4*p + 2)*2 + 1], fmaf(w1, wml[(4*p + 1)*2 + 1], w0 * wml[(4*p)*2 + 1])));
        }
    }
}

template <int WT, int NW>
static int launch_self_head(mi355x_ctx * ctx, const SHArgs & a, int H, double bytes, double flops) {
    if (a.n_kv <= 8*NW*2) return emit(ctx, "self_head", k_self_head<WT, NW, 2>, dim3((uint32_t) H), dim3(NW*64), 0, a, bytes, flops);
    if (a.n_kv <= 8*NW*4) return emit(ctx, "self_head", k_self_head<WT, NW, 4>, dim3((uint32_t) H), dim3(NW*64), 0, a, bytes, flops);
    return emit(ctx, "self_head", k_self_head<WT, NW, 64 / NW>, dim3((uint32_t) H), dim3(NW*64), 0, a, bytes, flops);
}

// d: the LayerNorm + Q / K / V description mi355x_gemv_fused takes (T = 1, three segments: seg[qi] the query projection with an F32
// destination or NULL, seg[ki] / seg[vi] writing this step's F16 rows into the caches); k / v / mask / scale: the operands of the
// flash_attn_ext that follows (k, v: [64, n_kv, H] views of the caches; the step's rows are key `new_key` of them).
// out: 128-key partial records as mi355x_flash_attn_partial leaves them.  MI355X_E_UNSUPPORTED: run the two launches.
extern "C" int mi355x_self_attn_head(mi355x_ctx * ctx, const mi355x_gemv_desc * d, int qi, int ki, int vi, const mi355x_tensor * k, const mi355x_tensor * v,
                                     const mi355x_tensor * mask, float scale, int new_key, mi355x_attn_partials * out) {
    static const bool on = !(getenv("GGML_MI355X_SELF_HEAD") && !atoi(getenv("GGML_MI355X_SELF_HEAD")));
    if (!on) return MI355X_E_UNSUPPORTED;
    if (d->T != 1 || d->nseg != 3 || !d->has_norm || !d->x || !d->ln_w || !d->ln_b || d->attn_part_o || d->x_planes || d->planes_out || d->cols) return MI355X_E_UNSUPPORTED;
    if (qi < 0 || ki < 0 || vi < 0 || qi > 2 || ki > 2 || vi > 2 || qi == ki || qi == vi || ki == vi) return MI355X_E_UNSUPPORTED;
    const int K = d->K, wt = d->seg[0].wtype;
    if (K < 128 || K > 2048 || K % 32 || ((uintptr_t) d->x | (uintptr_t) d->ln_w | (uintptr_t) d->ln_b) % 16) return MI355X_E_UNSUPPORTED;
    if (wt != MI355X_TYPE_Q4_0 && wt != MI355X_TYPE_Q5_0 && wt != MI355X_TYPE_Q8_0) return MI355X_E_UNSUPPORTED;
    if (k->type != MI355X_TYPE_F16 || v->type != MI355X_TYPE_F16 || k->ne[0] != 64 || v->ne[0] != 64 || k->nb[0] != 2 || v->nb[0] != 2) return MI355X_E_UNSUPPORTED;
    const int64_t n_kv = k->ne[1], H = k->ne[2];
    if (n_kv < 32 || n_kv > 512 || n_kv % 32 || v->ne[1] != n_kv || v->ne[2] != H || H < 1 || k->ne[3] != 1 || v->ne[3] != 1) return MI355X_E_UNSUPPORTED;
    if (k->nb[2] != 128 || v->nb[2] != 128 || k->nb[1] != H*128 || v->nb[1] != H*128) return MI355X_E_UNSUPPORTED;     // a key's heads side by side (whisper's caches)
    if (((uintptr_t) k->data | (uintptr_t) v->data) % 16) return MI355X_E_UNSUPPORTED;
    if (new_key < 0 || new_key >= n_kv) return MI355X_E_UNSUPPORTED;
    if (mask && (mask->type != MI355X_TYPE_F16 || mask->ne[0] < n_kv || mask->nb[0] != 2)) return MI355X_E_UNSUPPORTED;
    SHArgs a; memset(&a, 0, sizeof(a));
    const int order[3] = { qi, ki, vi };
    for (int s = 0; s < 3; s++) {
        const mi355x_gemv_seg & g = d->seg[order[s]];
        if (g.wtype != wt || g.N != H*64 || ((uintptr_t) g.w % 16) || g.ep.gelu || g.ep.residual || g.ep.bias_per_col) return MI355X_E_UNSUPPORTED;
        if (s == 0 ? (g.dst && g.dst_type != MI355X_TYPE_F32) : (!g.dst || g.dst_type != MI355X_TYPE_F16)) return MI355X_E_UNSUPPORTED;
        DGSeg & o = a.seg[s];
        o.w = g.w; o.N = g.N; o.nbt = (int64_t) g.N * (K / 32);
        o.bias = g.ep.bias; o.scale = g.ep.scale; o.has_scale = g.ep.has_scale; o.dst = g.dst; o.dst_f16 = s != 0;
    }
    // the step's rows must be key `new_key` of the views the attention reads
    if ((const char *) a.seg[1].dst != (const char *) k->data + (int64_t) new_key * k->nb[1] || (const char *) a.seg[2].dst != (const char *) v->data + (int64_t) new_key * v->nb[1]) return MI355X_E_UNSUPPORTED;
    a.x = d->x; a.K = K; a.eps = d->eps; a.ln_w = d->ln_w; a.ln_b = d->ln_b;
    a.kc = (const char *) k->data; a.vc = (const char *) v->data; a.k_nb1 = k->nb[1]; a.k_nb2 = k->nb[2]; a.v_nb1 = v->nb[1]; a.v_nb2 = v->nb[2];
    a.mask = mask ? (const char *) mask->data : nullptr; a.fa_scale = scale; a.n_kv = (int) n_kv; a.new_key = new_key;
    a.nparts = (int) ((n_kv + 127) / 128); a.valid = ctx->gelu_tab;
    mi355x_scratch_reset(ctx);
    const size_t nrec = (size_t) H * a.nparts;
    a.part_o  = (float *) mi355x_scratch_alloc(ctx, nrec * 64 * 4);
    a.part_ml = (float *) mi355x_scratch_alloc(ctx, nrec * 2 * 4);
    if (!a.part_o || !a.part_ml) return (int) hipErrorOutOfMemory;
    const double bytes = 3.0 * mi355x_type_row_bytes(wt, K) * H * 64 + 2.0 * n_kv * 64 * 2 * H + (double) K*4;
    const double flops = 2.0 * 3 * H * 64 * K + 4.0 * n_kv * 64 * H;
    int rc;
    switch (wt) {
        case MI355X_TYPE_Q4_0: rc = launch_self_head<MI355X_TYPE_Q4_0, 16>(ctx, a, (int) H, bytes, flops); break;
        case MI355X_TYPE_Q5_0: rc = launch_self_head<MI355X_TYPE_Q5_0, 16>(ctx, a, (int) H, bytes, flops); break;
        default:               rc = launch_self_head<MI355X_TYPE_Q8_0, 8>(ctx, a, (int) H, bytes, flops); break;
    }
    if (rc) return rc;
    out->part_o = a.part_o; out->part_ml = a.part_ml; out->nparts = a.nparts; out->T = 1; out->H = (int) H;
    return 0;
}
