// Probe for the ~30 us stall seen before the first chip-wide kernel of every decode step.
// Launches [small, wide, wide] after a host-idle period of D microseconds and lets rocprofv3 --kernel-trace show where
// the gap lands; mode 1 keeps one wave per CU spinning on a host flag during the idle period.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/stall_probe.hip -o gpurun_out/stall_probe
//   rocprofv3 --kernel-trace -f csv -d out -o p -- ./stall_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int TAG> __global__ void k_small(float * p) { p[threadIdx.x] += 1.0f; }
template <int TAG> __global__ void k_wide(float * p) { p[(size_t) blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }
__global__ void k_spin(volatile int * flag, int gen, long long cap_ticks) {
    long long t0 = __builtin_readcyclecounter();
    while (*flag < gen && __builtin_readcyclecounter() - t0 < cap_ticks) __builtin_amdgcn_s_sleep(8);
}

static void busy_wait_us(int us) {
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) {}
}

template <int TAG> static void run(hipStream_t s, hipStream_t s2, float * buf, volatile int * flag, int * dflag, int idle_us, int mode, int & gen) {
    for (int it = 0; it < 40; it++) {
        hipStreamSynchronize(s);
        if (mode == 1) { gen++; k_spin<<<256, 64, 0, s2>>>(dflag, gen, 200000000LL); }
        busy_wait_us(idle_us);
        k_small<TAG><<<1, 256, 0, s>>>(buf);
        k_wide<TAG><<<768, 320, 0, s>>>(buf);
        k_wide<TAG + 1000><<<768, 320, 0, s>>>(buf);
        k_small<TAG + 1000><<<1, 256, 0, s>>>(buf);
        k_wide<TAG + 2000><<<768, 320, 0, s>>>(buf);
        if (mode == 1) { *flag = gen; }
    }
    hipStreamSynchronize(s);
    if (mode == 1) hipStreamSynchronize(s2);
}

int main(int argc, char ** argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    float * buf; hipMalloc(&buf, 768 * 320 * 4); hipMemset(buf, 0, 768 * 320 * 4);
    int * hflag; hipHostMalloc(&hflag, 64, hipHostMallocMapped); *hflag = 0;
    int * dflag; hipHostGetDevicePointer((void **) &dflag, hflag, 0);
    int gen = 0;
    run<0>(s, s2, buf, hflag, dflag, 0, mode, gen);
    run<10>(s, s2, buf, hflag, dflag, 10, mode, gen);
    run<30>(s, s2, buf, hflag, dflag, 30, mode, gen);
    run<100>(s, s2, buf, hflag, dflag, 100, mode, gen);
    run<300>(s, s2, buf, hflag, dflag, 300, mode, gen);
    run<999>(s, s2, buf, hflag, dflag, 1000, mode, gen);
    printf("done mode %d\n", mode);
    return 0;
}
