// Persistent-layer experiment, second edition ("redo the probe so that it tests PREFETCH").
// A batch-1 decode step of large-v3 is 32 layers x 7 stages, each stage a mat-vec whose input is the WHOLE output of the previous
// one (all-to-all) and whose weights are 1-7 MB.  Ways to run that chain, all with the same arithmetic (int8 weights x per-32-block
// int8 activations, like the Q8_0 path) and bit-identical results:
//   A1  one kernel per stage, geometry of the first edition (256 workgroups x 4 waves, 5-20 sequential rows per wave), hipGraph;
//   A2  one kernel per stage, geometry of the PRODUCTION mat-vec (16-wave workgroups, one row per wave, every weight load of the
//       stage in flight before the activations are staged), eager launches and hipGraph — the baseline a persistent design has to beat;
//   B0  first edition's persistent launch: granule all-gather, weights loaded AFTER the gather (synchronisation only, no prefetch);
//   C   the design of MI355X_MICROARCH.md (rows prefetch-credit / gather-pass / engine-vs-launches): one workgroup per CU = 1 LOADER
//       wave + 3 CONSUMER waves.  The loader streams the workgroup's slice of the NEXT stage's weights (and of the read-only extra
//       stream) into an LDS ring with global_load_lds_dwordx4 (nt) while the consumers are still gathering / computing the current
//       stage, and waits with a COUNTED vmcnt so the run-ahead stays in flight across the barrier.  Consumers gather the input vector
//       as 8-byte {value, tag} granules (batched sc1 polls, no fences, no flags), quantize, take their rows from LDS, publish.
//       C0 = the same engine with the loader starting at the stage's own top (no run-ahead): isolates the prefetch credit.
// Every spin is bounded: a lost granule sets an error code instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/persist_probe.hip -o scripts/probes/_bin/persist_probe && scripts/probes/_bin/persist_probe [layers]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NWG 256
#define NT  256
#define NTC 192                  // consumer threads of the engine (waves 0..2), wave 3 loads
#define NSTAGE 7
#define EXTRA_ROW 6144           // bytes of the extra stream per output row of stage 3 (1280 x 6144 = 7.86 MB: one layer's cross K/V)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ constexpr int stage_n(int s) { return s == 0 ? 3840 : (s == 5 ? 5120 : 1280); }
__host__ __device__ constexpr int stage_k(int s) { return s == 1 ? 3840 : (s == 6 ? 5120 : 1280); }
__host__ __device__ constexpr bool stage_extra(int s) { return s == 3; }
__host__ __device__ constexpr size_t stage_woff(int s) { size_t o = 0; for (int i = 0; i < s; i++) o += (size_t) stage_n(i) * stage_k(i); return o; }
// stage 1 stands in for self-attention (reads a [1280 x 3840] weight: more bytes than the real K/V of <= 256 keys),
// stage 3 additionally streams the layer's cross K/V (7.86 MB): the Q-projection + cross-attention stage

struct Chain {
    const int8_t * w;            // all stage weights of ONE layer (re-used by every layer: the bytes streamed are what matters)
    const int8_t * extra;        // [1280][EXTRA_ROW]
    float * act[2];              // launches: ping-pong activation vectors (max 5120 floats)
    unsigned long long * gran;   // persistent: granules [NSTAGE][5120] {f32, tag}
    int * err;
    int n_layers;
};

// ---- shared arithmetic -----------------------------------------------------------------------------------------------
// 4 consecutive values of a 32-block held by each of 8 neighbouring lanes -> int8 + block scale (like quantize_row_q8_0)
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
template <int NTH> __device__ __forceinline__ void quant_from(const float * xv, int K, int8_t * xs, float * xd, int tid) {
    for (int base = tid * 4; base < ((K + NTH * 4 - 1) / (NTH * 4)) * NTH * 4; base += NTH * 4) {
        const bool in = base < K;
        quant4(in ? xv[base] : 0.0f, in ? xv[base + 1] : 0.0f, in ? xv[base + 2] : 0.0f, in ? xv[base + 3] : 0.0f, base, K, xs, xd);
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
// one row: lane b, b + 64, ... takes 32-byte blocks; w may point to global memory or LDS
__device__ __forceinline__ float row_dot(const int8_t * w, int K, const int8_t * xs, const float * xd, int lane) {
    float acc = 0.0f;
    for (int b = lane; b < K / 32; b += 64) {
        const int4 * wp = (const int4 *) (w + (size_t) b * 32);
        const int4 * xp = (const int4 *) (xs + b * 32);
        acc = fmaf(xd[b] * (1.0f / 64.0f), (float) dot32(wp[0], wp[1], xp[0], xp[1]), acc);
    }
    return wave_sum(acc);
}
// the read-only stream of a row (cross K/V stand-in): an order-independent integer checksum, by one wave
__device__ __forceinline__ float extra_row(const int8_t * e, int lane) {
    int s = 0;
    for (int i = lane * 16; i < EXTRA_ROW; i += 64 * 16) { const int4 v = *(const int4 *) (e + i); s += (v.x & 1) + (v.y & 1) + (v.z & 1) + (v.w & 1); }
    return (float) wave_sum_i(s) * 1e-9f;
}

// ---- A1: one launch per stage, 256 x 4 waves, sequential rows ----------------------------------------------------------
struct AArgs { Chain c; int stage; const float * x; float * y; };
__global__ void __launch_bounds__(NT) k_stage_a1(const AArgs a) {
    __shared__ __attribute__((aligned(16))) int8_t xs[5120];
    __shared__ float xd[160];
    const int s = a.stage, N = stage_n(s), K = stage_k(s);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wg = blockIdx.x;
    quant_from<NT>(a.x, K, xs, xd, tid);
    __syncthreads();
    const int per = N / NWG, r0 = wg * per;
    for (int r = r0 + wave; r < r0 + per; r += 4) {
        const float ex = stage_extra(s) ? extra_row(a.c.extra + (size_t) r * EXTRA_ROW, lane) : 0.0f;
        const float v = row_dot(a.c.w + stage_woff(s) + (size_t) r * K, K, xs, xd, lane);
        if (lane == 0) a.y[r] = tanhf(v + ex);
    }
}

// ---- A2: one launch per stage, production geometry: 16 waves, one row per wave, weights requested first ----------------
template <int KB, bool EXTRA> __global__ void __launch_bounds__(1024) k_stage_a2(const AArgs a) {
    __shared__ __attribute__((aligned(16))) int8_t xs[5120];
    __shared__ float xd[160];
    const int s = a.stage, K = stage_k(s);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = blockIdx.x * 16 + wave;
    const int8_t * w = a.c.w + stage_woff(s) + (size_t) r * K;
    int4 wr[KB][2];
#pragma unroll
    for (int i = 0; i < KB; i++) { const int b = lane + i * 64; if (b < K / 32) { wr[i][0] = ((const int4 *) (w + b * 32))[0]; wr[i][1] = ((const int4 *) (w + b * 32))[1]; } }
    int es = 0;
    if (EXTRA) { const int8_t * e = a.c.extra + (size_t) r * EXTRA_ROW;
#pragma unroll
        for (int i = 0; i < EXTRA_ROW / 1024; i++) { const int4 v = *(const int4 *) (e + lane * 16 + i * 1024); es += (v.x & 1) + (v.y & 1) + (v.z & 1) + (v.w & 1); } }
    __builtin_amdgcn_sched_barrier(0);
    quant_from<1024>(a.x, K, xs, xd, tid);
    __syncthreads();
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < KB; i++) { const int b = lane + i * 64; if (b < K / 32) { const int4 * xp = (const int4 *) (xs + b * 32); acc = fmaf(xd[b] * (1.0f / 64.0f), (float) dot32(wr[i][0], wr[i][1], xp[0], xp[1]), acc); } }
    acc = wave_sum(acc);
    const float ex = EXTRA ? (float) wave_sum_i(es) * 1e-9f : 0.0f;
    if (lane == 0) a.y[r] = tanhf(acc + ex);
}

// ---- B0: first edition's persistent kernel (no prefetch) ------------------------------------------------------------------
__device__ __forceinline__ void publish(unsigned long long * g, float v, unsigned tag) {
    const unsigned long long q = ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(v);
    __hip_atomic_store(g, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // one 8-byte write-through store: {value, tag}
}
__global__ void __launch_bounds__(NT) k_persistent_b0(const Chain c, const float * x0, float * y_out) {
    __shared__ __attribute__((aligned(16))) int8_t xs[5120];
    __shared__ float xd[160];
    __shared__ float xf[5120];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wg = blockIdx.x;
    unsigned tag = 1;
    for (int i = tid; i < stage_k(0); i += NT) xf[i] = x0[i];
    __syncthreads();
    for (int layer = 0; layer < c.n_layers; layer++) {
        for (int s = 0; s < NSTAGE; s++, tag++) {
            const int N = stage_n(s), K = stage_k(s);
            if (!(layer == 0 && s == 0)) {
                const unsigned long long * g = c.gran + (size_t) ((s + NSTAGE - 1) % NSTAGE) * 5120;
                const unsigned want = tag - 1;
                for (int i = tid; i < K; i += NT) {
                    unsigned long long q; int spins = 0;
                    do { q = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned) (q >> 32) != want && ++spins < (1 << 22));
                    if ((unsigned) (q >> 32) != want) *c.err = 1;                       // give up instead of hanging
                    xf[i] = __uint_as_float((unsigned) q);
                }
                __syncthreads();
            }
            quant_from<NT>(xf, K, xs, xd, tid);
            __syncthreads();
            const int per = N / NWG, r0 = wg * per;
            unsigned long long * go = c.gran + (size_t) s * 5120;
            for (int r = r0 + wave; r < r0 + per; r += 4) {
                const float ex = stage_extra(s) ? extra_row(c.extra + (size_t) r * EXTRA_ROW, lane) : 0.0f;
                const float v = row_dot(c.w + stage_woff(s) + (size_t) r * K, K, xs, xd, lane);
                if (lane == 0) { const float o = tanhf(v + ex); publish(go + r, o, tag); if (layer == c.n_layers - 1 && s == NSTAGE - 1) y_out[r] = o; }
            }
            __syncthreads();                                                            // xs / xf are rewritten by the next stage
        }
    }
}

// ---- C: loader wave + consumer waves ------------------------------------------------------------------------------------------
// LDS map (all LDS-DMA targets below 64 KB): a stage's slice (weights, then stage 3's 5 x 6144 extra bytes) never overlaps the slice of
// the stage before or after it, so stage t + 1 can land while stage t is read.  Slice bytes: 19200 19200 6400 6400+30720 6400 25600 25600.
__host__ __device__ constexpr int stage_lds(int s) { return s == 1 ? 19200 : (s == 3 ? 6400 : (s == 4 ? 43520 : (s == 6 ? 25600 : 0))); }
#define C_XS  51200
#define C_LDS (C_XS + 5120 + 160 * 4)

// 1 KiB per wave instruction: lane l's 16 bytes land at lds_dst + 16 l.  The statement owns M0.
__device__ __forceinline__ void glds16(const int8_t * gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int glds_span(const int8_t * g, unsigned lds, int bytes, int lane) {   // -> wave instructions issued
    const int n = (bytes + 1023) / 1024;
    for (int j = 0; j < n; j++) if (j * 1024 + lane * 16 < bytes) glds16(g + j * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds + j * 1024));
    return n;
}
__device__ __forceinline__ void wait_vm(int n) {           // s_waitcnt wants an immediate: the counts this chain can leave in flight
    switch (n) {
        case  7: asm volatile("s_waitcnt vmcnt(7)"  ::: "memory"); break;       //  5 x 1280
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;       // 15 x 1280
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;       // 20 x 1280, 5 x 5120
        case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;       //  5 x 1280 + 30 KB extra
        default: asm volatile("s_waitcnt vmcnt(0)"  ::: "memory"); break;
    }
}
__device__ __forceinline__ int stage_issue(const Chain & c, int s, int wg, unsigned lds0, int lane) {
    const int per = stage_n(s) / NWG, K = stage_k(s);
    int n = glds_span(c.w + stage_woff(s) + (size_t) wg * per * K, lds0 + stage_lds(s), per * K, lane);
    if (stage_extra(s)) n += glds_span(c.extra + (size_t) wg * per * EXTRA_ROW, lds0 + stage_lds(s) + per * K, per * EXTRA_ROW, lane);
    return n;
}

// gather K granules (tag `want`) and quantize them: thread t takes values [(g * 192 + t) * 4, +4), all of a pass's loads in flight together
template <int G> __device__ __forceinline__ void gather_quant(const unsigned long long * gr, int K, unsigned want, int ctid, int8_t * xs, float * xd, int * err, bool & failed) {
    unsigned long long q[G * 4];
    int spins = 0; bool ok;
    const int limit = failed ? 1 : (1 << 20);
    do {
#pragma unroll
        for (int g = 0; g < G; g++) { const int base = (g * NTC + ctid) * 4, at = base < K ? base : 0;      // past the end: re-read granule 0 (no predicated loads)
#pragma unroll
            for (int j = 0; j < 4; j++) q[g * 4 + j] = __hip_atomic_load(gr + at + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        ok = true;
#pragma unroll
        for (int i = 0; i < G * 4; i++) ok = ok && (unsigned) (q[i] >> 32) == want;
        if (!ok) __builtin_amdgcn_s_sleep(1);
    } while (!ok && ++spins < limit);
    if (!ok) { *err = 1; failed = true; }                                                // give up instead of hanging (and do not wait again)
#pragma unroll
    for (int g = 0; g < G; g++) { const int base = (g * NTC + ctid) * 4; const bool in = base < K;
        quant4(in ? __uint_as_float((unsigned) q[g * 4]) : 0.0f, in ? __uint_as_float((unsigned) q[g * 4 + 1]) : 0.0f, in ? __uint_as_float((unsigned) q[g * 4 + 2]) : 0.0f,
               in ? __uint_as_float((unsigned) q[g * 4 + 3]) : 0.0f, base, K, xs, xd); }
}

template <int AHEAD> __global__ void __launch_bounds__(NT) k_persistent_c(const Chain c, const float * x0, float * y_out) {
    extern __shared__ __attribute__((aligned(1024))) int8_t smem[];
    int8_t * xs   = smem + C_XS;
    float  * xd   = (float *) (xs + 5120);
    const unsigned lds0 = (unsigned) (unsigned long) (__attribute__((address_space(3))) int8_t *) smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wg = blockIdx.x;
    const int total = c.n_layers * NSTAGE;
    if (wave == 3) {
        // ---- loader ----
        if (AHEAD) { stage_issue(c, 0, wg, lds0, lane); }
        for (int t = 0; t < total; t++) {
            int in_flight_after = 0;
            if (AHEAD) { if (t + 1 < total) in_flight_after = stage_issue(c, (t + 1) % NSTAGE, wg, lds0, lane); }
            else       stage_issue(c, t % NSTAGE, wg, lds0, lane);
            wait_vm(in_flight_after);                       // stage t's bytes are in LDS, stage t + 1's stay in flight
            __builtin_amdgcn_s_barrier();                   // A: consumers may read the slot
            __builtin_amdgcn_s_barrier();                   // B: consumers are done with it
        }
        return;
    }
    // ---- consumers ----
    unsigned tag = 1;
    bool failed = false;
    for (int t = 0; t < total; t++, tag++) {
        const int s = t % NSTAGE, N = stage_n(s), K = stage_k(s);
        if (t == 0) quant_from<NTC>(x0, K, xs, xd, tid);
        else {
            const unsigned long long * gr = c.gran + (size_t) ((s + NSTAGE - 1) % NSTAGE) * 5120;
            if (K == 1280) gather_quant<2>(gr, K, tag - 1, tid, xs, xd, c.err, failed);
            else if (K == 3840) gather_quant<5>(gr, K, tag - 1, tid, xs, xd, c.err, failed);
            else gather_quant<7>(gr, K, tag - 1, tid, xs, xd, c.err, failed);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes done
        __builtin_amdgcn_s_barrier();                       // A
        const int per = N / NWG, r0 = wg * per;
        const int8_t * wl = smem + stage_lds(s), * el = wl + per * K;
        unsigned long long * go = c.gran + (size_t) s * 5120;
        for (int i = wave; i < per; i += 3) {
            const float ex = stage_extra(s) ? extra_row(el + i * EXTRA_ROW, lane) : 0.0f;
            const float v = row_dot(wl + i * K, K, xs, xd, lane);
            if (lane == 0) { const float o = tanhf(v + ex); publish(go + r0 + i, o, tag); if (t == total - 1) y_out[r0 + i] = o; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // B
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef void (*a2_fn)(const AArgs);
static a2_fn a2_kernel(int s) {
    if (stage_extra(s)) return k_stage_a2<1, true>;
    const int kb = (stage_k(s) / 32 + 63) / 64;
    return kb == 1 ? k_stage_a2<1, false> : (kb == 2 ? k_stage_a2<2, false> : k_stage_a2<3, false>);
}

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n_layers = argc > 1 ? atoi(argv[1]) : 32;
    Chain c; memset(&c, 0, sizeof(c));
    const size_t wbytes = stage_woff(NSTAGE), ebytes = (size_t) 1280 * EXTRA_ROW;
    c.n_layers = n_layers;
    std::vector<int8_t> hw(wbytes), he(ebytes);
    unsigned lcg = 12345;
    for (auto & v : hw) { lcg = lcg * 1664525u + 1013904223u; v = (int8_t) ((lcg >> 24) % 15 - 7); }
    for (auto & v : he) { lcg = lcg * 1664525u + 1013904223u; v = (int8_t) (lcg >> 24); }
    int8_t * dw, * de; CK(hipMalloc(&dw, wbytes)); CK(hipMalloc(&de, ebytes));
    CK(hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice)); CK(hipMemcpy(de, he.data(), ebytes, hipMemcpyHostToDevice));
    c.w = dw; c.extra = de;
    CK(hipMalloc(&c.act[0], 5120 * 4)); CK(hipMalloc(&c.act[1], 5120 * 4));
    CK(hipMalloc(&c.gran, (size_t) NSTAGE * 5120 * 8)); CK(hipMemset(c.gran, 0, (size_t) NSTAGE * 5120 * 8));
    CK(hipMalloc(&c.err, 4)); CK(hipMemset(c.err, 0, 4));
    std::vector<float> hx(1280);
    for (auto & v : hx) { lcg = lcg * 1664525u + 1013904223u; v = ((lcg >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float * dx0, * dy[6]; CK(hipMalloc(&dx0, 5120 * 4)); for (auto & p : dy) { CK(hipMalloc(&p, 5120 * 4)); CK(hipMemset(p, 0, 5120 * 4)); }
    CK(hipMemcpy(dx0, hx.data(), 1280 * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n_stage_total = n_layers * NSTAGE;
    printf("persistent-layer experiment: %d layers x %d stages, %.1f MB of int8 weights + %.2f MB extra stream per layer\n", n_layers, NSTAGE, wbytes / 1e6, ebytes / 1e6);

    auto report = [&](const char * name, double us, const float * y, const float * yref, int err) {
        std::vector<float> a(1280), b(1280);
        CK(hipMemcpy(a.data(), y, 1280 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), yref, 1280 * 4, hipMemcpyDeviceToHost));
        int same = 0; for (int i = 0; i < 1280; i++) same += memcmp(&a[i], &b[i], 4) == 0;
        printf("%-44s: %8.1f us per step = %6.2f us per layer = %5.2f us per stage   (identical to A1: %d / 1280, err %d)\n", name, us, us / n_layers, us / n_stage_total, same, err);
    };

    // ---- launch chains: A1 (graph), A2 (eager, graph) ----
    std::vector<AArgs> args((size_t) n_stage_total);
    auto fill_args = [&](float * yout) {
        for (int i = 0; i < n_stage_total; i++) { AArgs & a = args[i]; a.c = c; a.stage = i % NSTAGE; a.x = i == 0 ? dx0 : c.act[(i - 1) & 1]; a.y = i + 1 == n_stage_total ? yout : c.act[i & 1]; }
    };
    auto build_graph = [&](bool a2) {
        hipGraph_t g; hipGraphExec_t ge; CK(hipGraphCreate(&g, 0));
        std::vector<hipGraphNode_t> nodes((size_t) n_stage_total);
        for (int i = 0; i < n_stage_total; i++) {
            void * ka[1] = { &args[i] };
            hipKernelNodeParams p = {};
            const int s = i % NSTAGE;
            p.func = a2 ? (void *) a2_kernel(s) : (void *) k_stage_a1; p.gridDim = a2 ? dim3(stage_n(s) / 16) : dim3(NWG); p.blockDim = a2 ? dim3(1024) : dim3(NT); p.kernelParams = ka;
            CK(hipGraphAddKernelNode(&nodes[i], g, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &p));
        }
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        return ge;
    };
    auto time_it = [&](auto && fn) { double best = 1e30; for (int rep = 0; rep < 8; rep++) { CK(hipStreamSynchronize(st)); const double t0 = now_us(); fn(); CK(hipStreamSynchronize(st)); const double t = now_us() - t0; if (rep > 1 && t < best) best = t; } return best; };

    fill_args(dy[0]); hipGraphExec_t g1 = build_graph(false);
    const double tA1 = time_it([&] { CK(hipGraphLaunch(g1, st)); });
    report("A1 launches, first-edition geometry, graph", tA1, dy[0], dy[0], 0);
    fill_args(dy[1]); hipGraphExec_t g2 = build_graph(true);
    const double tA2g = time_it([&] { CK(hipGraphLaunch(g2, st)); });
    report("A2 launches, production geometry, graph", tA2g, dy[1], dy[0], 0);
    CK(hipMemsetAsync(dy[1], 0, 5120 * 4, st));
    const double tA2e = time_it([&] { for (int i = 0; i < n_stage_total; i++) { const int s = i % NSTAGE; a2_kernel(s)<<<dim3(stage_n(s) / 16), dim3(1024), 0, st>>>(args[i]); } });
    report("A2 launches, production geometry, eager", tA2e, dy[1], dy[0], 0);
    const double tA = tA2g < tA2e ? tA2g : tA2e;

    // ---- persistent launches ----
    CK(hipFuncSetAttribute((const void *) k_persistent_c<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C_LDS));
    CK(hipFuncSetAttribute((const void *) k_persistent_c<0>, hipFuncAttributeMaxDynamicSharedMemorySize, C_LDS));
    const char * names[3] = { "B0 persistent, weights after the gather", "C0 loader wave, no run-ahead", "C  loader wave, one stage of run-ahead" };
    double tP[3];
    for (int v = 0; v < 3; v++) {
        int herr = 0; double best = 1e30;
        for (int rep = 0; rep < 8; rep++) {
            CK(hipMemsetAsync(c.gran, 0, (size_t) NSTAGE * 5120 * 8, st));
            CK(hipStreamSynchronize(st));
            const double t0 = now_us();
            if (v == 0) k_persistent_b0<<<dim3(NWG), dim3(NT), 0, st>>>(c, dx0, dy[2]);
            else if (v == 1) k_persistent_c<0><<<dim3(NWG), dim3(NT), C_LDS, st>>>(c, dx0, dy[3]);
            else k_persistent_c<1><<<dim3(NWG), dim3(NT), C_LDS, st>>>(c, dx0, dy[4]);
            CK(hipStreamSynchronize(st));
            const double t = now_us() - t0;
            if (rep > 1 && t < best) best = t;
            CK(hipMemcpy(&herr, c.err, 4, hipMemcpyDeviceToHost));
            if (herr) { printf("%s: a granule never arrived (bounded spin gave up) — residency or visibility problem\n", names[v]); CK(hipMemset(c.err, 0, 4)); break; }
        }
        tP[v] = best;
        report(names[v], best, dy[2 + v], dy[0], herr);
    }
    printf("ratios to the better A2 chain (%.1f us): B0 %.2f, C0 %.2f, C %.2f;  C / A1 = %.2f;  prefetch credit C0 - C = %.2f us per stage\n",
           tA, tP[0] / tA, tP[1] / tA, tP[2] / tA, tP[2] / tA1, (tP[1] - tP[2]) / n_stage_total);
    printf("(production decode chain on the same GPU: 4.3 us per stage at 8 stages per layer, profiles/r03_step_trace_*.json)\n");
    return 0;
}
