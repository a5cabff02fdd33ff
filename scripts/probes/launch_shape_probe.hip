// Probe (round 6): what does the LAUNCH SHAPE of the merged chain's mat-vecs cost?  k_gemv_mx runs 80-320 workgroups of 512-1024 threads with 68-110 KB of
// dynamic LDS and a 2 KB argument struct; profiles/r05_mx_kernel_by_parts.txt measured 4.3-5.1 us for that kernel returning right after its prologue, while
// a trivial 256-thread kernel costs 1.8 us in a dependent chain (profiles/archive/r02_launch_probe.txt).  This program times dependent chains of a kernel
// that does one dependent global round trip (load -> store) for every combination of
//   block size {256, 512, 1024} x dynamic LDS {0, 64 KB, 110 KB} x argument bytes {64, 2048} x grid {80, 320},
// by hipEvents around 200 launches on one stream (device time per launch, launches issued back to back).
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/launch_shape_probe.hip -o scripts/probes/_bin/launch_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PAD> struct Args { float * p; int n; int touch_lds; long long pad[PAD]; int last; };

template <int PAD>
__global__ void k_probe(const Args<PAD> a) {
    extern __shared__ char lds[];
    float * p = a.p + ((size_t) blockIdx.x * blockDim.x + threadIdx.x) % a.n;
    float v = *p;
    if (a.touch_lds) { ((float *) lds)[threadIdx.x] = v; __syncthreads(); v = ((float *) lds)[(threadIdx.x + 1) % blockDim.x]; }
    *p = v + 1.0f + (float) a.last;
}

template <int PAD>
static void run(int threads, int lds, int grid, float * buf, int n, hipStream_t s) {
    Args<PAD> a = {}; a.p = buf; a.n = n; a.touch_lds = lds > 0; a.last = 0;
    CK(hipFuncSetAttribute((const void *) k_probe<PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_probe<PAD>, dim3(grid), dim3(threads), lds, s, a);
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_probe<PAD>, dim3(grid), dim3(threads), lds, s, a);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("threads %4d  lds %6d  args %4zu B  grid %3d : %6.2f us per dependent launch\n", threads, lds, sizeof(a), grid, best * 1e3 / 200);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 1 << 20;
    float * buf; CK(hipMalloc(&buf, (size_t) n * 4)); CK(hipMemset(buf, 0, (size_t) n * 4));
    for (int grid : { 80, 320 })
        for (int threads : { 256, 512, 1024 })
            for (int lds : { 0, 64 * 1024, 110 * 1024 }) {
                run<4>(threads, lds, grid, buf, n, s);
                run<252>(threads, lds, grid, buf, n, s);
            }
    return 0;
}
