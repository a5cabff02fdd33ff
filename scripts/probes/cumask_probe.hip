// Which bits of a hipExtStreamCreateWithCUMask mask select which XCD?  (needed for one CU-masked stream per XCD: VERDICT r02 next #1,
// "cheap experiment first").  For each candidate layout of "the 32 CUs of XCD x" a stream is created with that mask, a 512-block
// kernel records HW_REG_XCC_ID per block, and the histogram over XCDs is printed.  Also times a 1280-wave dependent launch chain on a
// masked stream against an unmasked one (what one XCD sustains alone).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/cumask_probe.hip -o scripts/probes/_bin/cumask_probe && scripts/probes/_bin/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(int * o) {
    unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) o[blockIdx.x] = (int) (v & 0xF);
}
__global__ void __launch_bounds__(320) k_chain(const float * in, float * out, int n) {       // a small dependent stage: every block reads all of `in`
    float s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 5 + (threadIdx.x >> 6)] = s * 1e-9f;
}

static void make_mask(uint32_t * m, int nwords, int layout, int xcd, int ncu) {
    memset(m, 0, nwords * 4);
    const int per = ncu / 8;
    for (int i = 0; i < per; i++) {
        const int bit = layout == 0 ? xcd + 8 * i        // strided: bit (8k + x) = k-th CU of XCD x
                                    : xcd * per + i;      // contiguous: bits [32x, 32x + 32)
        m[bit >> 5] |= 1u << (bit & 31);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount, nwords = (ncu + 31) / 32;
    printf("device: %s, %d CUs\n", p.name, ncu);
    int * d; CK(hipMalloc(&d, 512 * 4));
    std::vector<int> h(512);
    const char * names[2] = { "strided  (bit 8k+x)", "contiguous (bits 32x..32x+31)" };
    for (int layout = 0; layout < 2; layout++) for (int xcd : { 0, 3, 7 }) {
        std::vector<uint32_t> m(nwords);
        make_mask(m.data(), nwords, layout, xcd, ncu);
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t) nwords, m.data());
        if (e != hipSuccess) { printf("layout %s xcd %d: create failed: %s\n", names[layout], xcd, hipGetErrorString(e)); continue; }
        CK(hipMemsetAsync(d, 0xFF, 512 * 4, st));
        k_where<<<512, 64, 0, st>>>(d);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost));
        int hist[16] = { 0 };
        for (int v : h) if (v >= 0 && v < 16) hist[v]++;
        printf("layout %-30s target xcd %d -> blocks per XCC_ID:", names[layout], xcd);
        for (int i = 0; i < 8; i++) printf(" %d", hist[i]);
        printf("\n");
        CK(hipStreamDestroy(st));
    }
    // dependent chain of 260 launches x 256 blocks x 320 threads: unmasked stream vs one XCD (both layouts)
    float * a, * b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
    for (int mode = 0; mode < 3; mode++) {
        hipStream_t st;
        if (mode == 0) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        else { std::vector<uint32_t> m(nwords); make_mask(m.data(), nwords, mode - 1, 2, ncu); CK(hipExtStreamCreateWithCUMask(&st, (uint32_t) nwords, m.data())); }
        for (int rep = 0; rep < 3; rep++) {
            CK(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 260; i++) k_chain<<<256, 320, 0, st>>>(i & 1 ? b : a, i & 1 ? a : b, 1280);
            CK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("chain of 260 dependent 256x320 launches, %s: %.1f us (%.2f us per launch)\n",
                                 mode == 0 ? "all CUs" : (mode == 1 ? "one XCD, strided mask" : "one XCD, contiguous mask"), us, us / 260);
        }
        CK(hipStreamDestroy(st));
    }
    return 0;
}
