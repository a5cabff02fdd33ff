// What does a read-once stream of the vocabulary projection's size (45.6 MB) cost on this GPU when it is HBM-cold?  The ceiling for
// k_gemv8<6, 1, 8> (VERDICT r02 next #7 asks for 4.4 TB/s = 10.4 us).  8 distinct buffers (365 MB > the 256 MB Infinity Cache) are read
// round-robin by a trivial kernel: every thread sums `U` 16-byte loads issued back to back (nt), one store per wave.  Grid shapes from
// "everything in flight at once" to a few waves per CU walking the buffer.  hipEvent-bracketed single launches, median of 64.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/stream_probe.hip -o scripts/probes/_bin/stream_probe && scripts/probes/_bin/stream_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int U> __global__ void __launch_bounds__(256) k_read(const v4i * __restrict__ p, size_t n16, int * out) {
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    int s = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        v4i v[U];
        #pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p + i + u * stride);
        #pragma unroll
        for (int u = 0; u < U; u++) s += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const v4i v = __builtin_nontemporal_load(p + i); s += v.x ^ v.y ^ v.z ^ v.w; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t bytes = 45642080, n16 = bytes / 16;
    const int copies = 8;
    std::vector<v4i *> buf(copies);
    for (auto & b : buf) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
    int * out; CK(hipMalloc(&out, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("read-once stream of %.1f MB, %d buffers round-robin (HBM-cold)\n", bytes / 1e6, copies);
    const int grids[] = { 256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32, (int) ((n16 + 255) / 256) };
    for (int U : { 1, 4, 8 }) for (int g : grids) {
        std::vector<float> t;
        for (int r = 0; r < 80; r++) {
            const v4i * p = buf[r % copies];
            CK(hipEventRecord(e0, st));
            if (U == 1) k_read<1><<<g, 256, 0, st>>>(p, n16, out); else if (U == 4) k_read<4><<<g, 256, 0, st>>>(p, n16, out); else k_read<8><<<g, 256, 0, st>>>(p, n16, out);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 16) t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("U = %d loads in flight per thread, grid %6d x 256: median %6.2f us = %5.2f TB/s   (min %6.2f)\n", U, g, t[t.size() / 2], bytes / (t[t.size() / 2] * 1e-6) / 1e12, t[0]);
    }
    // the same events around an EMPTY launch: what the bracket itself costs
    std::vector<float> t;
    for (int r = 0; r < 80; r++) { CK(hipEventRecord(e0, st)); k_read<1><<<1, 64, 0, st>>>(buf[0], 0, out); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f); }
    std::sort(t.begin(), t.end());
    printf("empty kernel between the same events: median %.2f us\n", t[t.size() / 2]);
    return 0;
}
