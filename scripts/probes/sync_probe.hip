// How long after the last kernel of a decode step has finished does the host know?  ("spin on a pinned completion
// word written by the last kernel instead of hipStreamSynchronize".)  The same GPU work — a dependent chain of 40 kernels that each spin
// ~5 us, on a non-blocking stream — is waited for in five ways; the work is identical, so differences of the host-side wall time
// (enqueue of the first kernel -> wait returns, host already waiting when the chain ends) are differences of the completion latency.
//   a  hipStreamSynchronize
//   b  host spins on hipStreamQuery
//   c  a 1-thread kernel behind the chain writes a sequence number to pinned host memory, host spins on it
//   d  the LAST kernel of the chain writes the word itself (one workgroup: no arrival counting needed)
//   e  hipStreamWriteValue32 behind the chain, host spins on the word
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/sync_probe.hip -o scripts/probes/_bin/sync_probe && scripts/probes/_bin/sync_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(int * sink, long long ticks, volatile unsigned * flag, unsigned seq) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) { if (sink) *sink = (int) t0; if (flag) { __threadfence_system(); *flag = seq; } }
}
__global__ void k_signal(volatile unsigned * flag, unsigned seq) { *flag = seq; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int * sink; CK(hipMalloc(&sink, 4));
    unsigned * flag_h; CK(hipHostMalloc((void **) &flag_h, 64, hipHostMallocMapped));
    *flag_h = 0;
    unsigned * flag_d; CK(hipHostGetDevicePointer((void **) &flag_d, flag_h, 0));
    const long long ticks = 500;                   // wall_clock64 runs at 100 MHz: 5 us
    const int chain = 40, reps = 300;
    const char * names[5] = { "a hipStreamSynchronize", "b hipStreamQuery spin", "c signal kernel + pinned word", "d last kernel writes the word", "e hipStreamWriteValue32 + pinned word" };
    unsigned seq = 0;
    for (int mode = 0; mode < 5; mode++) {
        std::vector<double> t;
        bool ok = true;
        for (int r = 0; r < reps + 20 && ok; r++) {
            CK(hipStreamSynchronize(st));
            seq++;
            const double t0 = now_us();
            for (int i = 0; i < chain; i++) {
                const bool last = i + 1 == chain;
                k_spin<<<1, 64, 0, st>>>(sink, ticks, (mode == 3 && last) ? flag_d : nullptr, seq);
            }
            if (mode == 2) k_signal<<<1, 1, 0, st>>>(flag_d, seq);
            if (mode == 4) { hipError_t e = hipStreamWriteValue32(st, flag_d, seq, 0); if (e != hipSuccess) { printf("%s: %s\n", names[mode], hipGetErrorString(e)); (void) hipGetLastError(); ok = false; break; } }
            if (mode == 0) CK(hipStreamSynchronize(st));
            else if (mode == 1) { while (hipStreamQuery(st) == hipErrorNotReady) { } }
            else { const double lim = now_us() + 1e6; while (*(volatile unsigned *) flag_h != seq && now_us() < lim) { } if (*(volatile unsigned *) flag_h != seq) { printf("%s: the word never arrived\n", names[mode]); ok = false; } }
            const double dt = now_us() - t0;
            if (r >= 20) t.push_back(dt);
        }
        if (!ok || t.empty()) continue;
        std::sort(t.begin(), t.end());
        printf("%-40s median %8.2f us   p10 %8.2f   p90 %8.2f   (chain of %d x ~5 us kernels)\n", names[mode], t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], chain);
    }
    CK(hipStreamSynchronize(st));
    return 0;
}
