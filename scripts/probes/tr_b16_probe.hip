// Probe: what does gfx950's ds_read_b64_tr_b16 deliver?  LDS holds u16 element i at byte 2i; lane l reads from byte address l*8 (4 elements
// of its own), and from a [4 rows][16 cols] block with a 128-byte row pitch (the V tile of the attention kernel).  Prints, per lane, the four
// elements it received.   hipcc --offload-arch=gfx950 -O2 scripts/probes/tr_b16_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void k_probe(unsigned short * out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short) i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned ad;
    if (mode == 0) ad = (unsigned) (size_t) lds + l * 8;                                   // lane l supplies elements 4l .. 4l+3
    else ad = (unsigned) (size_t) lds + ((l & 15) >> 2) * 128 + (l & 3) * 8 + (l >> 4) * 512;   // 16-lane group g: rows 4g..4g+3 (pitch 64 elements), lane -> row (l&15)/4, cols 4(l&3)..
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
    out[l*4 + 0] = (unsigned short) (v.x & 0xFFFF); out[l*4 + 1] = (unsigned short) (v.x >> 16);
    out[l*4 + 2] = (unsigned short) (v.y & 0xFFFF); out[l*4 + 3] = (unsigned short) (v.y >> 16);
}

int main() {
    unsigned short * d; CK(hipMalloc(&d, 64 * 4 * 2));
    unsigned short h[256];
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("mode %d (%s)\n", mode, mode == 0 ? "lane l supplies elements 4l..4l+3" : "row pitch 64 elements: lane -> (row (l&15)/4 + 4(l>>4), cols 4(l&3)..)");
        for (int l = 0; l < 64; l++) {
            printf("  lane %2d: %4d %4d %4d %4d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
            if (mode == 1) printf("   = (row,col) (%d,%d) (%d,%d) (%d,%d) (%d,%d)", h[l*4] / 64, h[l*4] % 64, h[l*4+1] / 64, h[l*4+1] % 64, h[l*4+2] / 64, h[l*4+2] % 64, h[l*4+3] / 64, h[l*4+3] % 64);
            printf("\n");
        }
    }
    return 0;
}
