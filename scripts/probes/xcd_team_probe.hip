// XCD-local teams: the last untested single-column design.
// A batch-1 decode step is 32 layers x 7 dependent stages (mat-vecs whose input is the whole output of the previous one).  The launch
// chain pays ~4.06 us per stage whatever the bytes (profiles/r03b_kernel_trace_summary_*.txt); the device-wide persistent kernels of
// scripts/probes/persist_probe.hip lost to it because their all-to-all exchange crosses the XCDs (per-XCD L2s are not coherent: every granule
// is a fabric write, every poll a fabric read).  Here the GPU is cut along its L2s instead:
//   * ONE launch of 256 workgroups (one per CU, 8 waves); a workgroup reads HW_REG_XCC_ID and joins the TEAM of its XCD (32 CUs);
//   * every team runs ONE stream's chain on its own: a stage's rows are split over the team's 32 workgroups, the activation vector is
//     exchanged as 8-byte {value, tag} granules THROUGH THE XCD'S OWN L2 — plain stores keep the line in that L2, `sc1` loads bypass the
//     reader's L1 and are served by it (MI355X_MICROARCH.md, visibility table) — no agent-scope fence, no cross-XCD traffic, no lock-step
//     between teams: 8 independent streams, each at 1/8 of the machine;
//   * the first weight rows of the NEXT stage are requested before a wave starts polling for the current stage's input.
// Same arithmetic as persist_probe's chains (int8 weights x per-32-block int8 activations, tanh, stage 3 also streams 7.86 MB of
// per-stream "cross K/V"), results checked bit for bit against a launch chain.  Every spin is bounded.
// Findings (profiles/r04_xcd_team_probe.txt): exchange alone 1.8 us per stage; rows requested after the exchange: 9.7-10.3 us per stage whatever the
// team count or bytes; next stage's first batch requested BEFORE the exchange (k_teams3): 7.53 us = 4740 tokens/s with 8 teams = 1.98 x the merged chains.
// Question to answer with a number: us per stage per team with all 8 teams running, against the 4.06 us launch chain that serves ONE
// stream (or 8 columns in ~2.5 ms per step when merged: 9.35 chunks/s for 8 streams).  8 teams at S us per stage are
// 8 / (224 * S us) tokens per second; break-even with twice the merged chains' 9.35 chunks/s (4790 tokens/s) is S = 7.5 us.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/xcd_team_probe.hip -o scripts/probes/_bin/xcd_team_probe && scripts/probes/_bin/xcd_team_probe [layers] [sixteenths of a weight row read: 11]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NXCD 8
#define TEAM 32                  // workgroups (CUs) per team
#define NT   1024
#define NW   (NT / 64)
#define NTT  512                 // threads of a team workgroup (8 waves: two rows of weights in registers per wave without spilling)
#define NWT  (NTT / 64)
#define NSTAGE 7
#define EXTRA_ROW 6144           // bytes of the per-stream extra stream per output row of stage 3 (1280 x 6144 = 7.86 MB: one layer's cross K/V)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ constexpr int stage_n(int s) { return s == 0 ? 3840 : (s == 5 ? 5120 : 1280); }
__host__ __device__ constexpr int stage_k(int s) { return s == 1 ? 3840 : (s == 6 ? 5120 : 1280); }
__host__ __device__ constexpr bool stage_extra(int s) { return s == 3; }
__host__ __device__ constexpr size_t stage_woff(int s) { size_t o = 0; for (int i = 0; i < s; i++) o += (size_t) stage_n(i) * stage_k(i); return o; }

struct Chain {
    const int8_t * w;            // all stage weights of ONE layer (re-used by every layer: the bytes streamed are what matters)
    const int8_t * extra;        // [NXCD][1280][EXTRA_ROW]: every stream has its own
    unsigned long long * gran;   // [NXCD][NSTAGE][5120] {f32, tag}
    int * team_count;            // [NXCD]
    int * err;
    int n_layers, active_teams, sc1_stores, prefetch;
    int rows_mode;               // 0: one row at a time (first edition); 1: batches of rows, two batches in flight
    int anatomy;                 // 0: the real thing; 1: no exchange (a stage does not wait for its input: loads + arithmetic only); 2: exchange only (no weight rows)
    int frac16;                  // sixteenths of a row's bytes that are read: 16 = int8 weights, 11 = the bytes of Q5_0 (0.6875 B per weight)
};

__device__ __forceinline__ void quant4(float v0, float v1, float v2, float v3, int base, int K, int8_t * xs, float * xd) {
    float amax = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64)); amax = fmaxf(amax, __shfl_xor(amax, 2, 64)); amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
    if (base < K) {
        char4 q; q.x = (int8_t) rintf(v0 * id); q.y = (int8_t) rintf(v1 * id); q.z = (int8_t) rintf(v2 * id); q.w = (int8_t) rintf(v3 * id);
        *(char4 *) (xs + base) = q;
        if ((base & 31) == 0) xd[base >> 5] = amax / 127.0f;
    }
}
__device__ __forceinline__ int dot32(const int4 w0, const int4 w1, const int4 x0, const int4 x1) {
    int s = 0;
    s = __builtin_amdgcn_sdot4(w0.x, x0.x, s, false); s = __builtin_amdgcn_sdot4(w0.y, x0.y, s, false);
    s = __builtin_amdgcn_sdot4(w0.z, x0.z, s, false); s = __builtin_amdgcn_sdot4(w0.w, x0.w, s, false);
    s = __builtin_amdgcn_sdot4(w1.x, x1.x, s, false); s = __builtin_amdgcn_sdot4(w1.y, x1.y, s, false);
    s = __builtin_amdgcn_sdot4(w1.z, x1.z, s, false); s = __builtin_amdgcn_sdot4(w1.w, x1.w, s, false);
    return s;
}
__device__ __forceinline__ float wave_sum(float acc) { for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64); return acc; }
__device__ __forceinline__ int wave_sum_i(int acc) { for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64); return acc; }

// a row's weight blocks (lane b, b + 64, b + 128 take 32-byte blocks; clamped duplicates are dropped in the dot) and, for stage 3, its extra stream
struct Row { int4 w[3][2]; int4 e[6]; };
__device__ __forceinline__ int eff_blocks(int K, int frac16) { return ((K >> 5) * frac16 + 15) >> 4; }
__device__ __forceinline__ void row_load(Row & r, const int8_t * w, int K, const int8_t * e, int lane, int frac16) {
    const int nb = eff_blocks(K, frac16);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i * 64 < nb) { const int b = lane + i * 64, bc = b < nb ? b : nb - 1; r.w[i][0] = ((const int4 *) (w + (size_t) bc * 32))[0]; r.w[i][1] = ((const int4 *) (w + (size_t) bc * 32))[1]; }
    }
    if (e) {
#pragma unroll
        for (int i = 0; i < EXTRA_ROW / 1024; i++) r.e[i] = *(const int4 *) (e + lane * 16 + i * 1024);
    }
}
__device__ __forceinline__ float row_value(const Row & r, int K, bool extra, const int8_t * xs, const float * xd, int lane, int frac16) {
    const int nb = eff_blocks(K, frac16);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int b = lane + i * 64;
        if (i * 64 < nb && b < nb) { const int4 * xp = (const int4 *) (xs + b * 32); acc = fmaf(xd[b] * (1.0f / 64.0f), (float) dot32(r.w[i][0], r.w[i][1], xp[0], xp[1]), acc); }
    }
    acc = wave_sum(acc);
    float ex = 0.0f;
    if (extra) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < EXTRA_ROW / 1024; i++) s += (r.e[i].x & 1) + (r.e[i].y & 1) + (r.e[i].z & 1) + (r.e[i].w & 1);
        ex = (float) wave_sum_i(s) * 1e-9f;
    }
    return tanhf(acc + ex);
}

// ---- reference: one launch per stage (16 waves, one row per wave) ------------------------------------------------------------------
struct AArgs { const int8_t * w; const int8_t * extra; int stage; const float * x; float * y; int frac16; };
__global__ void __launch_bounds__(NT) k_stage(const AArgs a) {
    __shared__ __attribute__((aligned(16))) int8_t xs[5120];
    __shared__ float xd[160];
    const int s = a.stage, K = stage_k(s);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = blockIdx.x * NW + wave;
    Row row;
    row_load(row, a.w + stage_woff(s) + (size_t) r * K, K, stage_extra(s) ? a.extra + (size_t) r * EXTRA_ROW : nullptr, lane, a.frac16);
    __builtin_amdgcn_sched_barrier(0);
    for (int base = tid * 4; base < ((K + NT * 4 - 1) / (NT * 4)) * NT * 4; base += NT * 4) {
        const bool in = base < K;
        quant4(in ? a.x[base] : 0.0f, in ? a.x[base + 1] : 0.0f, in ? a.x[base + 2] : 0.0f, in ? a.x[base + 3] : 0.0f, base, K, xs, xd);
    }
    __syncthreads();
    const float v = row_value(row, K, stage_extra(s), xs, xd, lane, a.frac16);
    if (lane == 0) a.y[r] = v;
}

// ---- the teams ------------------------------------------------------------------------------------------------------------------------
// four consecutive granules (32 bytes), L1-bypassing: served by the XCD's L2
__device__ __forceinline__ void load_gran4(const unsigned long long * p, unsigned long long q[4]) {
    int4 a, b;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    q[0] = ((unsigned long long) (unsigned) a.y << 32) | (unsigned) a.x; q[1] = ((unsigned long long) (unsigned) a.w << 32) | (unsigned) a.z;
    q[2] = ((unsigned long long) (unsigned) b.y << 32) | (unsigned) b.x; q[3] = ((unsigned long long) (unsigned) b.w << 32) | (unsigned) b.z;
}
__device__ __forceinline__ void store_gran(unsigned long long * p, unsigned long long v, bool sc1) {
    if (sc1) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else     asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");          // plain: the line stays in this XCD's L2
}

// ---- rows in BATCHES (team kernel, rows_mode 1): a wave requests RB rows at once and, before it reduces them, the next RB rows — in
// straight-line code, so that the compiler's counted vmcnt waits leave the younger batch in flight (the row-at-a-time loop above ends up
// behind a vmcnt(0) per row: one memory round trip per row, 1.2 us — what the first edition of this probe measured as "the team").
template <int KB, bool EXTRA> struct RowT { int4 w[KB][2]; int4 e[EXTRA ? 6 : 1]; };
template <int KB, bool EXTRA> __device__ __forceinline__ void rowt_load(RowT<KB, EXTRA> & r, const int8_t * w, int nb, const int8_t * e, int lane) {
#pragma unroll
    for (int i = 0; i < KB; i++) { const int b = lane + i * 64, bc = b < nb ? b : nb - 1; r.w[i][0] = ((const int4 *) (w + (size_t) bc * 32))[0]; r.w[i][1] = ((const int4 *) (w + (size_t) bc * 32))[1]; }
    if (EXTRA) {
#pragma unroll
        for (int i = 0; i < EXTRA_ROW / 1024; i++) r.e[i] = *(const int4 *) (e + lane * 16 + i * 1024);
    }
}
template <int KB, bool EXTRA> __device__ __forceinline__ float rowt_value(const RowT<KB, EXTRA> & r, int nb, const int8_t * xs, const float * xd, int lane) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < KB; i++) {
        const int b = lane + i * 64;
        if (b < nb) { const int4 * xp = (const int4 *) (xs + b * 32); acc = fmaf(xd[b] * (1.0f / 64.0f), (float) dot32(r.w[i][0], r.w[i][1], xp[0], xp[1]), acc); }
    }
    acc = wave_sum(acc);
    float ex = 0.0f;
    if (EXTRA) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < EXTRA_ROW / 1024; i++) s += (r.e[i].x & 1) + (r.e[i].y & 1) + (r.e[i].z & 1) + (r.e[i].w & 1);
        ex = (float) wave_sum_i(s) * 1e-9f;
    }
    return tanhf(acc + ex);
}
struct StageCtx { const int8_t * wbase; const int8_t * extra; int K, nb, per, r0, wave, lane; const int8_t * xs; const float * xd; unsigned long long * go; unsigned tag; int sc1; float * y_last; };
template <int KB, bool EXTRA, int RB> __device__ __forceinline__ void batch_load(RowT<KB, EXTRA> * R, const StageCtx & q, int b) {
#pragma unroll
    for (int j = 0; j < RB; j++) {
        int i = q.wave + NWT * (b * RB + j); i = i < q.per ? i : q.per - 1;            // past the end: the last row again (never published)
        const int row = q.r0 + i;
        rowt_load<KB, EXTRA>(R[j], q.wbase + (size_t) row * q.K, q.nb, EXTRA ? q.extra + (size_t) row * EXTRA_ROW : nullptr, q.lane);
    }
}
template <int KB, bool EXTRA, int RB> __device__ __forceinline__ void batch_compute(const RowT<KB, EXTRA> * R, const StageCtx & q, int b) {
#pragma unroll
    for (int j = 0; j < RB; j++) {
        const int i = q.wave + NWT * (b * RB + j);
        const float v = rowt_value<KB, EXTRA>(R[j], q.nb, q.xs, q.xd, q.lane);
        if (q.lane == 0 && i < q.per) {
            store_gran(q.go + q.r0 + i, ((unsigned long long) q.tag << 32) | (unsigned long long) __float_as_uint(v), q.sc1 != 0);
            if (q.y_last) q.y_last[q.r0 + i] = v;
        }
    }
}
template <int KB, bool EXTRA, int RB> __device__ __forceinline__ void stage_rows(const StageCtx & q) {
    const int mine = q.wave < q.per ? (q.per - q.wave + NWT - 1) / NWT : 0, nbatch = (mine + RB - 1) / RB;
    RowT<KB, EXTRA> A[RB], B[RB];
    if (nbatch > 0) batch_load<KB, EXTRA, RB>(A, q, 0);
    for (int b = 0; b < nbatch; b += 2) {
        if (b + 1 < nbatch) batch_load<KB, EXTRA, RB>(B, q, b + 1);
        batch_compute<KB, EXTRA, RB>(A, q, b);
        if (b + 2 < nbatch) batch_load<KB, EXTRA, RB>(A, q, b + 2);
        if (b + 1 < nbatch) batch_compute<KB, EXTRA, RB>(B, q, b + 1);
    }
}


// ---- third edition: stages known at compile time, the NEXT stage's first batch of rows requested before the exchange ---------------------
// A stage's weights do not depend on its input: the first batch of stage t + 1 is requested while stage t's last batch is still being
// reduced, and lands while the team exchanges stage t's results (1.8 us) — the one memory round trip per stage that the editions above pay
// after every exchange.  Register buffers are generic (16 x int4 each: X0 / X1 alternate as "first batch of this stage" / "first batch of the
// next stage", Y is the stage's own second buffer); a row of type (KB, EX) takes 2 KB + 6 EX of them.
#ifndef GBUF
#define GBUF 16
#endif
template <int FR, int S> struct StT {
    static constexpr int K = stage_k(S), N = stage_n(S), NB = ((K >> 5) * FR + 15) >> 4, KB = (NB + 63) >> 6, PER = N / TEAM;
    static constexpr bool EX = stage_extra(S);
    static constexpr int RW = 2 * KB + (EX ? 6 : 0), RB = GBUF / RW;                                    // int4 per row, rows per batch (8 / 4 / 2 / 2)
    static constexpr int MINE_MAX = (PER + NWT - 1) / NWT, NBATCH = (MINE_MAX + RB - 1) / RB;      // (the same for every wave up to clamped duplicates)
};
struct TeamCtx { const Chain * c; const int8_t * extra; unsigned long long * gran; const float * x0; float * y_team; int8_t * xs; float * xd; int rank, wave, lane, tid; bool failed; int zero; };

template <int KB, bool EX, int RB> __device__ __forceinline__ void gbatch_load(int4 (&R)[GBUF], const StageCtx & q, int b) {
    constexpr int RW = 2 * KB + (EX ? 6 : 0);
#pragma unroll
    for (int j = 0; j < RB; j++) {
        int i = q.wave + NWT * (b * RB + j); i = i < q.per ? i : q.per - 1;            // past the end: the last row again (never published)
        const int row = q.r0 + i;
        const int8_t * w = q.wbase + (size_t) row * q.K;
#pragma unroll
        for (int k = 0; k < KB; k++) { const int blk = q.lane + k * 64, bc = blk < q.nb ? blk : q.nb - 1; R[j*RW + 2*k] = ((const int4 *) (w + (size_t) bc * 32))[0]; R[j*RW + 2*k + 1] = ((const int4 *) (w + (size_t) bc * 32))[1]; }
        if (EX) {
            const int8_t * e = q.extra + (size_t) row * EXTRA_ROW;
#pragma unroll
            for (int k = 0; k < EXTRA_ROW / 1024; k++) R[j*RW + 2*KB + k] = *(const int4 *) (e + q.lane * 16 + k * 1024);
        }
    }
}
template <int KB, bool EX, int RB> __device__ __forceinline__ void gbatch_compute(const int4 (&R)[GBUF], const StageCtx & q, int b) {
    constexpr int RW = 2 * KB + (EX ? 6 : 0);
#pragma unroll
    for (int j = 0; j < RB; j++) {
        const int i = q.wave + NWT * (b * RB + j);
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int blk = q.lane + k * 64;
            if (blk < q.nb) { const int4 * xp = (const int4 *) (q.xs + blk * 32); acc = fmaf(q.xd[blk] * (1.0f / 64.0f), (float) dot32(R[j*RW + 2*k], R[j*RW + 2*k + 1], xp[0], xp[1]), acc); }
        }
        acc = wave_sum(acc);
        float ex = 0.0f;
        if (EX) {
            int sm = 0;
#pragma unroll
            for (int k = 0; k < EXTRA_ROW / 1024; k++) { const int4 e = R[j*RW + 2*KB + k]; sm += (e.x & 1) + (e.y & 1) + (e.z & 1) + (e.w & 1); }
            ex = (float) wave_sum_i(sm) * 1e-9f;
        }
        const float v = tanhf(acc + ex);
        if (q.lane == 0 && i < q.per) {
            store_gran(q.go + q.r0 + i, ((unsigned long long) q.tag << 32) | (unsigned long long) __float_as_uint(v), q.sc1 != 0);
            if (q.y_last) q.y_last[q.r0 + i] = v;
        }
    }
}
template <int FR, int S> __device__ __forceinline__ StageCtx stage_ctx(const TeamCtx & T, unsigned tag, bool last) {
    typedef StT<FR, S> P;
    StageCtx q = { T.c->w + stage_woff(S) + T.zero, T.extra + T.zero, P::K, P::NB, P::PER, T.rank * P::PER, T.wave, T.lane, T.xs + T.zero, T.xd + T.zero, T.gran + (size_t) S * 5120 + T.zero, tag, T.c->sc1_stores, last ? T.y_team : nullptr };
    return q;
}
// input of stage S: x0 (very first stage) or the granules of stage S - 1 with tag `want`
template <int FR, int S> __device__ __forceinline__ void stage_input(TeamCtx & T, bool first, unsigned want) {
    constexpr int K = StT<FR, S>::K;
    const unsigned long long * gr = T.gran + (size_t) ((S + NSTAGE - 1) % NSTAGE) * 5120 + T.zero;
    for (int base = T.tid * 4; base < ((K + NTT * 4 - 1) / (NTT * 4)) * NTT * 4; base += NTT * 4) {
        const bool in = base < K;
        float v[4] = { 0, 0, 0, 0 };
        if (in) {
            if (first) { v[0] = T.x0[base]; v[1] = T.x0[base + 1]; v[2] = T.x0[base + 2]; v[3] = T.x0[base + 3]; }
            else {
                unsigned long long q[4]; int spins = 0; bool ok;
                const int limit = T.failed ? 1 : (1 << 18);
                do {
                    load_gran4(gr + base, q);
                    ok = T.c->anatomy == 1 || ((unsigned) (q[0] >> 32) == want && (unsigned) (q[1] >> 32) == want && (unsigned) (q[2] >> 32) == want && (unsigned) (q[3] >> 32) == want);
                    if (!ok) __builtin_amdgcn_s_sleep(1);
                } while (!ok && ++spins < limit);
                if (!ok) { T.failed = true; atomicExch(T.c->err, 1); }
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = __uint_as_float((unsigned) q[j]);
            }
        }
        quant4(v[0], v[1], v[2], v[3], base, K, T.xs, T.xd);
    }
}
// one stage: batches 0, 1, 2, ... of the stage alternate between the two register buffers, batch 0 (requested during the previous stage) sits
// in `A`; the next stage's first batch is requested into whichever buffer does NOT hold this stage's last batch, before that batch is reduced
template <int FR, int S> __device__ __forceinline__ void run_stage(TeamCtx & T, int4 (&A)[GBUF], int4 (&Bf)[GBUF], bool first, bool last, bool more, unsigned tag) {
    typedef StT<FR, S> P; typedef StT<FR, (S + 1) % NSTAGE> Pn;
    static_assert(P::NBATCH >= 1 && P::NBATCH <= 3, "batches per stage");
    asm volatile("" : "+s"(T.zero));                      // (keeps this stage's address arithmetic inside the stage: see k_teams3)
    stage_input<FR, S>(T, first, tag - 1);
    __syncthreads();
    const StageCtx q = stage_ctx<FR, S>(T, tag, last);
    const StageCtx qn = stage_ctx<FR, (S + 1) % NSTAGE>(T, tag + 1, false);
    if constexpr (P::NBATCH > 1) gbatch_load<P::KB, P::EX, P::RB>(Bf, q, 1); else { if (more) gbatch_load<Pn::KB, Pn::EX, Pn::RB>(Bf, qn, 0); }
    gbatch_compute<P::KB, P::EX, P::RB>(A, q, 0);
    if constexpr (P::NBATCH > 1) {
        if constexpr (P::NBATCH > 2) gbatch_load<P::KB, P::EX, P::RB>(A, q, 2); else { if (more) gbatch_load<Pn::KB, Pn::EX, Pn::RB>(A, qn, 0); }
        gbatch_compute<P::KB, P::EX, P::RB>(Bf, q, 1);
    }
    if constexpr (P::NBATCH > 2) {
        if (more) gbatch_load<Pn::KB, Pn::EX, Pn::RB>(Bf, qn, 0);
        gbatch_compute<P::KB, P::EX, P::RB>(A, q, 2);
    }
    __syncthreads();                                       // xs / xd are rewritten by the next stage
}
// the seven stages of a layer; PAR = which buffer holds the stage's first batch (it flips after a stage with an odd number of batches)
template <int FR, int S, int PAR> struct LayerRun {
    static constexpr int NEXT = PAR ^ (StT<FR, S>::NBATCH & 1);
    static constexpr int END = LayerRun<FR, S + 1, NEXT>::END;
    static __device__ __forceinline__ void go(TeamCtx & T, int4 (&X0)[GBUF], int4 (&X1)[GBUF], bool first_layer, bool last_layer, unsigned & tag) {
        if constexpr (PAR) run_stage<FR, S>(T, X1, X0, first_layer && S == 0, false, true, tag); else run_stage<FR, S>(T, X0, X1, first_layer && S == 0, false, true, tag);
        tag++;
        LayerRun<FR, S + 1, NEXT>::go(T, X0, X1, first_layer, last_layer, tag);
    }
};
template <int FR, int PAR> struct LayerRun<FR, NSTAGE - 1, PAR> {
    static constexpr int END = PAR ^ (StT<FR, NSTAGE - 1>::NBATCH & 1);
    static __device__ __forceinline__ void go(TeamCtx & T, int4 (&X0)[GBUF], int4 (&X1)[GBUF], bool, bool last_layer, unsigned & tag) {
        if constexpr (PAR) run_stage<FR, NSTAGE - 1>(T, X1, X0, false, last_layer, !last_layer, tag); else run_stage<FR, NSTAGE - 1>(T, X0, X1, false, last_layer, !last_layer, tag);
        tag++;
    }
};
template <int FR> __global__ void __launch_bounds__(NTT) k_teams3(const Chain c, const float * x0, float * y_out) {
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    __shared__ int s_rank, s_xcc;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        s_xcc = (int) (v & 7);
        s_rank = atomicAdd(c.team_count + s_xcc, 1);
    }
    __syncthreads();
    const int xcc = s_xcc, rank = s_rank;
    if (rank >= TEAM) { if (tid == 0) atomicExch(c.err, 2); return; }
    if (xcc >= c.active_teams) return;
    TeamCtx T = { &c, c.extra + (size_t) xcc * 1280 * EXTRA_ROW, c.gran + (size_t) xcc * NSTAGE * 5120, x0, y_out + (size_t) xcc * 1280, smem, (float *) (smem + 5120), rank, tid >> 6, tid & 63, tid, false, 0 };
    int4 X0[GBUF], X1[GBUF];
    { const StageCtx q0 = stage_ctx<FR, 0>(T, 1, false); gbatch_load<StT<FR, 0>::KB, StT<FR, 0>::EX, StT<FR, 0>::RB>(X0, q0, 0); }
    unsigned tag = 1;
    // a layer may end with the buffers' roles swapped (LayerRun<FR, 0, 0>::END == 1): two layers per trip bring them back (n_layers is even)
    constexpr int LP = LayerRun<FR, 0, 0>::END;
    for (int layer = 0; layer < c.n_layers; layer += 2) {
        // (an opaque zero in every weight address: left alone, the compiler hoists the ~200 loop-invariant row addresses of the 14 unrolled stages
        //  out of this loop and spills them)
        asm volatile("" : "+s"(T.zero));
        LayerRun<FR, 0, 0>::go(T, X0, X1, layer == 0, false, tag);
        asm volatile("" : "+s"(T.zero));
        LayerRun<FR, 0, LP>::go(T, X0, X1, false, layer + 2 >= c.n_layers, tag);
    }
}

template <int RM> __global__ void __launch_bounds__(NTT) k_teams(const Chain c, const float * x0, float * y_out) {
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];      // (sized so that one workgroup fits per CU)
    int8_t * xs = smem;
    float * xd = (float *) (smem + 5120);
    __shared__ int s_rank, s_xcc;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        s_xcc = (int) (v & 7);
        s_rank = atomicAdd(c.team_count + s_xcc, 1);
    }
    __syncthreads();
    const int xcc = s_xcc, rank = s_rank;
    if (rank >= TEAM) { if (tid == 0) atomicExch(c.err, 2); return; }          // placement was not 32 per XCD: reported, not hung
    if (xcc >= c.active_teams) return;
    unsigned long long * gran = c.gran + (size_t) xcc * NSTAGE * 5120;
    const int8_t * extra = c.extra + (size_t) xcc * 1280 * EXTRA_ROW;
    const int total = c.n_layers * NSTAGE;
    unsigned tag = 1;
    bool failed = false;
    Row cur, nxt;              // (RM == 0 only)
    auto first_row = [&](int s, Row & r) {
        const int per = stage_n(s) / TEAM, K = stage_k(s), i = wave;
        if (i < per) { const int row = rank * per + i; row_load(r, c.w + stage_woff(s) + (size_t) row * K, K, stage_extra(s) ? extra + (size_t) row * EXTRA_ROW : nullptr, lane, c.frac16); }
    };
    if (RM == 0) first_row(0, cur);
    for (int t = 0; t < total; t++, tag++) {
        const int s = t % NSTAGE, N = stage_n(s), K = stage_k(s);
        // ---- input vector: x0 or the previous stage's granules of THIS team ----
        if (t == 0) {
            for (int base = tid * 4; base < ((K + NTT * 4 - 1) / (NTT * 4)) * NTT * 4; base += NTT * 4) {
                const bool in = base < K;
                quant4(in ? x0[base] : 0.0f, in ? x0[base + 1] : 0.0f, in ? x0[base + 2] : 0.0f, in ? x0[base + 3] : 0.0f, base, K, xs, xd);
            }
        } else {
            const unsigned long long * gr = gran + (size_t) ((s + NSTAGE - 1) % NSTAGE) * 5120;
            const unsigned want = tag - 1;
            for (int base = tid * 4; base < ((K + NTT * 4 - 1) / (NTT * 4)) * NTT * 4; base += NTT * 4) {
                const bool in = base < K;
                float v[4] = { 0, 0, 0, 0 };
                if (in) {
                    unsigned long long q[4]; int spins = 0; bool ok;
                    const int limit = failed ? 1 : (1 << 18);
                    do {
                        load_gran4(gr + base, q);
                        ok = c.anatomy == 1 || ((unsigned) (q[0] >> 32) == want && (unsigned) (q[1] >> 32) == want && (unsigned) (q[2] >> 32) == want && (unsigned) (q[3] >> 32) == want);
                        if (!ok) __builtin_amdgcn_s_sleep(1);
                    } while (!ok && ++spins < limit);
                    if (!ok) { failed = true; atomicExch(c.err, 1); }
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = __uint_as_float((unsigned) q[j]);
                }
                quant4(v[0], v[1], v[2], v[3], base, K, xs, xd);
            }
        }
        __syncthreads();
        // ---- this workgroup's rows: wave w takes rows w, w + 16, ...; the next row is in flight while the current one is reduced ----
        const int per = N / TEAM, r0 = rank * per;
        unsigned long long * go = gran + (size_t) s * 5120;
        const int8_t * wbase = c.w + stage_woff(s);
        if (c.anatomy == 2) {          // exchange only: publish this workgroup's granules without touching the weights
            for (int i = tid; i < per; i += NTT) store_gran(go + r0 + i, ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(0.25f), c.sc1_stores != 0);
        } else if constexpr (RM == 1) {
            StageCtx q = { wbase, extra, K, eff_blocks(K, c.frac16), per, r0, wave, lane, xs, xd, go, tag, c.sc1_stores, t == total - 1 ? y_out + (size_t) xcc * 1280 : nullptr };
            const int kb = (q.nb + 63) >> 6;
            if (stage_extra(s))  stage_rows<1, true, 2>(q);
            else if (kb == 1)    stage_rows<1, false, 8>(q);
            else if (kb == 2)    stage_rows<2, false, 4>(q);
            else                 stage_rows<3, false, 2>(q);
        } else if constexpr (RM == 0)
        for (int i = wave; i < per; i += NWT) {
            const bool more = i + NWT < per;
            if (more) { const int row = r0 + i + NWT; row_load(nxt, wbase + (size_t) row * K, K, stage_extra(s) ? extra + (size_t) row * EXTRA_ROW : nullptr, lane, c.frac16); }
            else if (c.prefetch && t + 1 < total) first_row((t + 1) % NSTAGE, nxt);          // the next stage's first row: in flight across the exchange
            const float v = row_value(cur, K, stage_extra(s), xs, xd, lane, c.frac16);
            if (lane == 0) {
                store_gran(go + r0 + i, ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(v), c.sc1_stores != 0);
                if (t == total - 1) y_out[(size_t) xcc * 1280 + r0 + i] = v;
            }
            cur = nxt;
        }
        if (RM == 0 && !c.prefetch && c.anatomy != 2 && t + 1 < total) first_row((t + 1) % NSTAGE, cur);
        __syncthreads();                                       // xs / xd are rewritten by the next stage
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n_layers = argc > 1 ? atoi(argv[1]) : 32;
    const int frac16 = argc > 2 ? atoi(argv[2]) : 11;
    Chain c; memset(&c, 0, sizeof(c));
    c.frac16 = frac16;
    const size_t wbytes = stage_woff(NSTAGE), ebytes = (size_t) 1280 * EXTRA_ROW;
    c.n_layers = n_layers;
    std::vector<int8_t> hw(wbytes), he(ebytes);
    unsigned lcg = 12345;
    for (auto & v : hw) { lcg = lcg * 1664525u + 1013904223u; v = (int8_t) ((lcg >> 24) % 15 - 7); }
    for (auto & v : he) { lcg = lcg * 1664525u + 1013904223u; v = (int8_t) (lcg >> 24); }
    int8_t * dw, * de; CK(hipMalloc(&dw, wbytes)); CK(hipMalloc(&de, ebytes * NXCD));
    CK(hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice));
    for (int x = 0; x < NXCD; x++) CK(hipMemcpy(de + (size_t) x * ebytes, he.data(), ebytes, hipMemcpyHostToDevice));      // own copy per stream (same values: one reference serves all)
    c.w = dw; c.extra = de;
    CK(hipMalloc(&c.gran, (size_t) NXCD * NSTAGE * 5120 * 8));
    CK(hipMalloc(&c.team_count, NXCD * 4)); CK(hipMalloc(&c.err, 4)); CK(hipMemset(c.err, 0, 4));
    std::vector<float> hx(1280);
    for (auto & v : hx) { lcg = lcg * 1664525u + 1013904223u; v = ((lcg >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float * dx0, * act[2], * yref, * yteam;
    CK(hipMalloc(&dx0, 5120 * 4)); CK(hipMalloc(&act[0], 5120 * 4)); CK(hipMalloc(&act[1], 5120 * 4)); CK(hipMalloc(&yref, 5120 * 4)); CK(hipMalloc(&yteam, NXCD * 1280 * 4));
    CK(hipMemcpy(dx0, hx.data(), 1280 * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int total = n_layers * NSTAGE;
    double layer_bytes = 0; for (int s = 0; s < NSTAGE; s++) layer_bytes += (double) stage_n(s) * (((stage_k(s) >> 5) * frac16 + 15) / 16) * 32; layer_bytes += (double) ebytes;
    printf("XCD-local teams: %d layers x %d stages, %d/16 of every int8 weight row read (11/16 = the bytes of Q5_0): %.1f MB of weights + %.2f MB per-stream extra per layer = %.1f MB per token (large-v3 Q5_0: 802)\n",
           n_layers, NSTAGE, frac16, (layer_bytes - ebytes) / 1e6, ebytes / 1e6, layer_bytes * n_layers / 1e6);

    // ---- reference launch chain (eager, production geometry) ----
    auto chain = [&]() {
        for (int i = 0; i < total; i++) {
            const int s = i % NSTAGE;
            AArgs a = { dw, de, s, i == 0 ? dx0 : act[(i - 1) & 1], i + 1 == total ? yref : act[i & 1], frac16 };
            k_stage<<<dim3(stage_n(s) / NW), dim3(NT), 0, st>>>(a);
        }
    };
    double tA = 1e30;
    for (int rep = 0; rep < 6; rep++) { CK(hipStreamSynchronize(st)); const double t0 = now_us(); chain(); CK(hipStreamSynchronize(st)); const double t = now_us() - t0; if (rep > 1 && t < tA) tA = t; }
    printf("%-64s: %8.1f us per token = %5.2f us per stage  (one stream on the whole GPU)\n", "launch chain, 16-wave workgroups, one row per wave, eager", tA, tA / total);
    std::vector<float> ref(1280); CK(hipMemcpy(ref.data(), yref, 1280 * 4, hipMemcpyDeviceToHost));

    // ---- teams ----
    const size_t lds = 96 * 1024;                                  // > half of the CU's 160 KB: one workgroup per CU
    CK(hipFuncSetAttribute((const void *) k_teams<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    CK(hipFuncSetAttribute((const void *) k_teams<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    CK(hipFuncSetAttribute((const void *) k_teams3<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    CK(hipFuncSetAttribute((const void *) k_teams3<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    struct V { const char * name; int teams, sc1, prefetch, anatomy, rows_mode; } vs[] = {
        { "8 teams, compile-time stages, next stage's first batch prefetched", 8, 0, 0, 0, 2 },
        { "4 teams, compile-time stages + prefetch",                      4, 0, 0, 0, 2 },
        { "1 team,  compile-time stages + prefetch",                      1, 0, 0, 0, 2 },
        { "8 teams, compile-time stages + prefetch, sc1 stores",          8, 1, 0, 0, 2 },
        { "anatomy: 8 teams, compile-time stages + prefetch, NO exchange", 8, 0, 0, 1, 2 },
        { "8 teams, rows in batches (two batches in flight)",             8, 0, 0, 0, 1 },
        { "8 teams, rows in batches, sc1 stores",                         8, 1, 0, 0, 1 },
        { "4 teams, rows in batches",                                     4, 0, 0, 0, 1 },
        { "1 team alone, rows in batches",                                1, 0, 0, 0, 1 },
        { "anatomy: 8 teams, rows in batches, NO exchange",               8, 0, 0, 1, 1 },
        { "first edition (one row at a time): 8 teams, next stage prefetched", 8, 0, 1, 0 },
        { "8 teams, plain stores (XCD-local L2), no prefetch",           8, 0, 0, 0 },
        { "8 teams, sc1 stores (what a device-wide exchange needs)",     8, 1, 1, 0 },
        { "1 team alone (the other 7 XCDs idle), prefetch",              1, 0, 1, 0 },
        { "4 teams, prefetch",                                           4, 0, 1, 0 },
        { "anatomy: 8 teams, NO exchange (loads + arithmetic only)",      8, 0, 0, 1 },
        { "anatomy: 8 teams, exchange ONLY (no weight rows)",             8, 0, 0, 2 },
        { "anatomy: 1 team, NO exchange",                                 1, 0, 0, 1 },
    };
    for (const V & v : vs) {
        c.active_teams = v.teams; c.sc1_stores = v.sc1; c.prefetch = v.prefetch; c.anatomy = v.anatomy; c.rows_mode = v.rows_mode;
        double best = 1e30; int herr = 0; int counts[NXCD];
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemsetAsync(c.gran, 0, (size_t) NXCD * NSTAGE * 5120 * 8, st));
            CK(hipMemsetAsync(c.team_count, 0, NXCD * 4, st));
            CK(hipMemsetAsync(yteam, 0, NXCD * 1280 * 4, st));
            CK(hipStreamSynchronize(st));
            const double t0 = now_us();
            if (v.rows_mode == 2) { if (frac16 == 16) k_teams3<16><<<dim3(NXCD * TEAM), dim3(NTT), lds, st>>>(c, dx0, yteam); else k_teams3<11><<<dim3(NXCD * TEAM), dim3(NTT), lds, st>>>(c, dx0, yteam); }
            else if (v.rows_mode) k_teams<1><<<dim3(NXCD * TEAM), dim3(NTT), lds, st>>>(c, dx0, yteam);
            else             k_teams<0><<<dim3(NXCD * TEAM), dim3(NTT), lds, st>>>(c, dx0, yteam);
            CK(hipStreamSynchronize(st));
            const double t = now_us() - t0;
            if (rep > 1 && t < best) best = t;
            CK(hipMemcpy(&herr, c.err, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(counts, c.team_count, NXCD * 4, hipMemcpyDeviceToHost));
            if (herr) { CK(hipMemset(c.err, 0, 4)); break; }
        }
        std::vector<float> got((size_t) NXCD * 1280); CK(hipMemcpy(got.data(), yteam, got.size() * 4, hipMemcpyDeviceToHost));
        int same = 0; for (int x = 0; x < v.teams; x++) for (int i = 0; i < 1280; i++) same += memcmp(&got[(size_t) x * 1280 + i], &ref[i], 4) == 0;
        const double tok_s = v.teams / (best * 1e-6);
        printf("%-64s: %8.1f us per token per team = %5.2f us per stage; %7.0f tokens/s aggregate = %5.2f x the chain; identical %d / %d, err %d, blocks per XCD %d %d %d %d %d %d %d %d\n",
               v.name, best, best / total, tok_s, tok_s / (1e6 / tA), same, v.teams * 1280, herr, counts[0], counts[1], counts[2], counts[3], counts[4], counts[5], counts[6], counts[7]);
        if (herr == 1) printf("    (a granule never arrived: bounded spin gave up)\n");
        if (herr == 2) printf("    (placement was not 32 workgroups per XCD)\n");
    }
    printf("reference points: production decode chain 4.06 us per stage, 1 stream (profiles/r03b_kernel_trace_summary_*.txt); merged chains 8 streams 9.35 chunks/s = 2394 tokens/s;\n"
           "break-even with 2 x the merged chains (VERDICT r03 next #6) = 4790 tokens/s aggregate = 7.5 us per stage per team with 8 teams.\n");
    return 0;
}
