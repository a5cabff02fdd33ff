// Probe: what does ONE dependent launch cost on this stack, and where do the kernel arguments live?
// A decode step is ~230 dependent launches of 1-4 MB each; round 1 measured 4.7 us for a 1-workgroup kernel inside the
// hipGraph while MI355X_MICROARCH.md prices a dependent boundary at 1.45 us.  This program measures, under the process
// environment it is started with (HIP_FORCE_DEV_KERNARG, DEBUG_CLR_GRAPH_PACKET_CAPTURE, ...):
//   * chain of N dependent launches, eager and as one hipGraph (built node by node like the backend does), for
//     {1, 256, 1280} workgroups and {16, 384}-byte argument structs, and a chain of "pointer only" kernels whose
//     parameters sit in a device-resident block;
//   * inside the kernel: s_memtime at the first instruction vs. after the LAST dword of the argument struct has
//     arrived (kernarg fetch latency), and after one dependent global load.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/launch_probe.hip -o scripts/probes/_bin/launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Small { float * p; unsigned long long * stamps; };
struct Big   { float * p; unsigned long long * stamps; long long pad[44]; int last; int pad2; };    // 16 + 352 + 8 = 376 bytes
static_assert(sizeof(Big) >= 376, "");

template <typename A> __device__ __forceinline__ int last_of(const A & a);
template <> __device__ __forceinline__ int last_of<Small>(const Small & a) { return (int) (size_t) a.stamps & 0; }
template <> __device__ __forceinline__ int last_of<Big>(const Big & a) { return a.last; }

template <typename A>
__global__ void __launch_bounds__(320) k_chain(const A a) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int last = last_of<A>(a);                       // forces the tail of the kernarg segment to be fetched
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(last));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float * p = a.p + (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const float v = *p;                                    // one dependent global round trip
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(v));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    *p = v + 1.0f + (float) last;
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) { a.stamps[0] = t0; a.stamps[1] = t1; a.stamps[2] = t2; }
}

// parameters in a device-resident block: the kernarg segment holds one pointer
__global__ void __launch_bounds__(320) k_indirect(const Big * __restrict__ blk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const Big a = *blk;
    const int last = a.last;
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(last));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float * p = a.p + (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const float v = *p;
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(v));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    *p = v + 1.0f + (float) last;
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) { a.stamps[0] = t0; a.stamps[1] = t1; a.stamps[2] = t2; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename A>
static void run_case(const char * label, const void * func, const A & args, int grid, int nchain, hipStream_t s, unsigned long long * stamps_d) {
    void * kargs[1] = { (void *) &args };
    // eager
    for (int i = 0; i < 64; i++) CK(hipLaunchKernel(func, dim3(grid), dim3(320), kargs, 0, s));
    CK(hipStreamSynchronize(s));
    double best_e = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        const double t0 = now_us();
        for (int i = 0; i < nchain; i++) CK(hipLaunchKernel(func, dim3(grid), dim3(320), kargs, 0, s));
        CK(hipStreamSynchronize(s));
        const double t = (now_us() - t0) / nchain;
        if (t < best_e) best_e = t;
    }
    // graph, node by node with explicit dependencies (what the backend does)
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> nodes(nchain);
    for (int i = 0; i < nchain; i++) {
        hipKernelNodeParams p = {};
        p.func = (void *) func; p.gridDim = dim3(grid); p.blockDim = dim3(320); p.sharedMemBytes = 0; p.kernelParams = kargs; p.extra = nullptr;
        CK(hipGraphAddKernelNode(&nodes[i], g, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &p));
    }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double best_g = 1e30, best_ev = 1e30;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 8; rep++) {
        const double t0 = now_us();
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        const double t = (now_us() - t0) / nchain;
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (t < best_g) best_g = t;
        if (ms * 1e3 / nchain < best_ev) best_ev = ms * 1e3 / nchain;
    }
    unsigned long long st[3] = { 0, 0, 0 };
    CK(hipMemcpy(st, stamps_d, sizeof(st), hipMemcpyDeviceToHost));
    // s_memtime counts shader cycles (MI355X_MICROARCH.md): printed raw and as us at 2.1 GHz
    printf("%-34s grid %5d  eager %6.2f us/launch   graph %6.2f us/launch (wall) %6.2f (events)   in-kernel: kernarg %6llu ticks (%.2f us), +load %6llu ticks (%.2f us)\n",
           label, grid, best_e, best_g, best_ev, st[1] - st[0], (double) (st[1] - st[0]) / 2100.0, st[2] - st[1], (double) (st[2] - st[1]) / 2100.0);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    const char * names[] = { "HIP_FORCE_DEV_KERNARG", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DEBUG_HIP_KERNARG_COPY_OPT", "ROC_USE_FGS_KERNARG", "DEBUG_HIP_GRAPH_BATCH_SIZE", "GPU_MAX_HW_QUEUES" };
    printf("env:");
    for (const char * n : names) printf(" %s=%s", n, getenv(n) ? getenv(n) : "(unset)");
    printf("\n");
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float * buf; CK(hipMalloc(&buf, (size_t) 1280 * 320 * 4)); CK(hipMemset(buf, 0, (size_t) 1280 * 320 * 4));
    unsigned long long * stamps; CK(hipMalloc(&stamps, 64)); CK(hipMemset(stamps, 0, 64));
    Small sm = { buf, stamps };
    Big bg = {}; bg.p = buf; bg.stamps = stamps; bg.last = 0;
    Big * blk; CK(hipMalloc(&blk, sizeof(Big))); CK(hipMemcpy(blk, &bg, sizeof(Big), hipMemcpyHostToDevice));
    const int nchain = 230;
    for (int grid : { 1, 256, 1280 }) {
        run_case("16-byte args", (const void *) k_chain<Small>, sm, grid, nchain, s, stamps);
        run_case("376-byte args", (const void *) k_chain<Big>, bg, grid, nchain, s, stamps);
        run_case("pointer to device-resident args", (const void *) k_indirect, blk, grid, nchain, s, stamps);
    }
    return 0;
}
