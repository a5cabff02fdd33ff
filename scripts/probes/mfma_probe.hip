// Probe: what does ONE K-step of the ring GEMM's inner loop cost, piece by piece?  (r03: the 128 x 128 tile spends ~1.2 us per
// 64-deep K-step on 16 MFMAs that take 0.21 us; profiles/r03b_gemm_dbg.txt shows the LDS-DMA and the compute part are each that
// slow on their own.)  One 256-thread workgroup per CU (grid 256) or two (grid 512), N iterations of a body, wall time by hipEvents
// and s_memtime ticks of wave 0.  Bodies (NT = 2: wave tile 64 x 64, 16 MFMAs + 16 ds_read_b128 per step; NT = 1: 64 x 32, 8 + 12):
//   0  MFMAs only (fragments stay in registers)
//   1  fragment reads per 16-deep slice right before their MFMAs, one register set (what hipcc made of the old loop)
//   2  all reads of the step up front, then the MFMAs (counted waits by inline asm)
//   3  body 2 + s_barrier per step
//   4  body 3 + the step's LDS-DMA (global_load_lds, 2-stage ring: vmcnt(0) before the barrier)
//   5  body 3 + LDS-DMA with a 4-stage ring (vmcnt(2 stages))
//   6  s_barrier only
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int lds_off(int row, int slot) { return row*128 + ((slot ^ ((row >> 1) & 7)) << 4); }

struct Args { const char * src; float * out; unsigned long long * ticks; int iters; long long src_stride; };

template <int BODY, int NT>
__global__ void __launch_bounds__(256) k_probe(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int BN = NT * 64;
    constexpr int STAGE = (128 + BN) * 128;
    constexpr int NST = BODY == 5 ? 4 : 2;
    constexpr int G = (128 + BN) / 32;                 // LDS-DMA instructions per wave and stage
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < NST * STAGE / 4; i += 256) ((float *) lds)[i] = 0.001f * (float) (i & 63);
    __syncthreads();
    floatx16 acc[2][NT];
    for (int i = 0; i < 2; i++) for (int j = 0; j < NT; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    half8_t af[4][2], bf[4][NT];
    for (int kk = 0; kk < 4; kk++) {
        for (int i = 0; i < 2; i++) for (int e = 0; e < 8; e++) af[kk][i][e] = (_Float16) (0.01f * (lane + e + kk));
        for (int j = 0; j < NT; j++) for (int e = 0; e < 8; e++) bf[kk][j][e] = (_Float16) (0.02f * (lane - e + j));
    }
    const char * g = a.src + (long long) blockIdx.x * a.src_stride + (long long) wave * G * 1024 + lane * 16;
    auto issue = [&](int st, int step) {
        #pragma unroll
        for (int i = 0; i < G; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g + (long long) step * (G * 4 * 1024) + i * 1024),
                                             (__attribute__((address_space(3))) void *) (lds + st * STAGE + (wave * G + i) * 1024), 16, 0, 0);
    };
    if (BODY == 4 || BODY == 5) { for (int p = 0; p < NST - 1; p++) issue(p, p); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < a.iters; it++) {
        const char * ldsA = lds + (it % NST) * STAGE;
        const char * ldsB = ldsA + 128*128;
        if (BODY == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (BODY == 5) { if (G == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        if (BODY >= 3) __builtin_amdgcn_s_barrier();
        if (BODY == 4 || BODY == 5) issue((it + NST - 1) % NST, (it + NST - 1) & 15);
        if (BODY == 6) continue;
        if (BODY == 0) {
            #pragma unroll
            for (int kk = 0; kk < 4; kk++)
                #pragma unroll
                for (int i = 0; i < 2; i++)
                    #pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
        } else if (BODY == 1) {
            #pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                half8_t xa[2], xb[NT];
                const int slot = kk*2 + (lane >> 5);
                #pragma unroll
                for (int i = 0; i < 2; i++) xa[i] = *(const half8_t *) (ldsA + lds_off(wm*64 + i*32 + (lane & 31), slot));
                #pragma unroll
                for (int j = 0; j < NT; j++) xb[j] = *(const half8_t *) (ldsB + lds_off(wn*(BN/2) + j*32 + (lane & 31), slot));
                #pragma unroll
                for (int i = 0; i < 2; i++)
                    #pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[i], xb[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // all reads up front through inline asm (hipcc does not count them), counted waits per slice
            #pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int slot = kk*2 + (lane >> 5);
                #pragma unroll
                for (int i = 0; i < 2; i++) {
                    const unsigned ad = (unsigned) (size_t) (ldsA + lds_off(wm*64 + i*32 + (lane & 31), slot));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(af[kk][i]) : "v"(ad));
                }
                #pragma unroll
                for (int j = 0; j < NT; j++) {
                    const unsigned ad = (unsigned) (size_t) (ldsB + lds_off(wn*(BN/2) + j*32 + (lane & 31), slot));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bf[kk][j]) : "v"(ad));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            #pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                if (NT == 2) {
                    if (kk == 0) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); else if (kk == 1) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                    else if (kk == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    if (kk == 0) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); else if (kk == 1) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                    else if (kk == 2) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                #pragma unroll
                for (int i = 0; i < 2; i++)
                    #pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < NT; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    a.out[(size_t) blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) { a.ticks[0] = t1 - t0; }
}

template <int BODY, int NT>
static void run(const char * label, int grid, const Args & a0, hipStream_t st) {
    constexpr int NST = BODY == 5 ? 4 : 2;
    const size_t lds = (size_t) NST * (128 + NT * 64) * 128;
    CK(hipFuncSetAttribute((const void *) k_probe<BODY, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    Args a = a0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<BODY, NT>), dim3(grid), dim3(256), lds, st, a);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_probe<BODY, NT>), dim3(grid), dim3(256), lds, st, a);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long ticks = 0; CK(hipMemcpy(&ticks, a.ticks, 8, hipMemcpyDeviceToHost));
    const double us_step = ms * 1e3 / a.iters, tick_step = (double) ticks / a.iters;
    const double mfma_cyc = (NT == 2 ? 16 : 8) * 32.0;
    printf("%-58s grid %3d NT %d: %7.3f us per step (wall)  %8.1f s_memtime ticks per step  (its MFMAs alone: %4.0f cycles)\n", label, grid, NT, us_step, tick_step, BODY == 6 ? 0.0 : mfma_cyc);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const long long stride = 16 * 32 * 1024;                      // 16 distinct stages of up to 32 KB per workgroup
    char * src; CK(hipMalloc(&src, (size_t) stride * 768)); CK(hipMemset(src, 0x11, (size_t) stride * 768));
    float * out; CK(hipMalloc(&out, 768 * 256 * 4));
    unsigned long long * ticks; CK(hipMalloc(&ticks, 64));
    Args a = { src, out, ticks, 2000, stride };
    for (int grid : { 256, 512 }) {
        run<0, 2>("0 MFMAs only", grid, a, st);
        run<1, 2>("1 reads per slice right before its MFMAs", grid, a, st);
        run<2, 2>("2 all reads up front, counted waits", grid, a, st);
        run<3, 2>("3 = 2 + s_barrier", grid, a, st);
        run<4, 2>("4 = 3 + LDS-DMA, 2-stage ring (vmcnt(0))", grid, a, st);
        run<5, 2>("5 = 3 + LDS-DMA, 4-stage ring (vmcnt(2 stages))", grid, a, st);
        run<6, 2>("6 s_barrier only", grid, a, st);
    }
    for (int grid : { 256, 768 }) {
        run<0, 1>("0 MFMAs only", grid, a, st);
        run<1, 1>("1 reads per slice right before its MFMAs", grid, a, st);
        run<2, 1>("2 all reads up front, counted waits", grid, a, st);
        run<3, 1>("3 = 2 + s_barrier", grid, a, st);
        run<4, 1>("4 = 3 + LDS-DMA, 2-stage ring (vmcnt(0))", grid, a, st);
        run<5, 1>("5 = 3 + LDS-DMA, 4-stage ring (vmcnt(2 stages))", grid, a, st);
    }
    return 0;
}
