// Hardware probe for the assumptions mmq.hip makes about gfx950 (run once on the GPU box; prints PASS / FAIL per item):
//  1. v_mfma_i32_32x32x32_i8: lane l supplies row (l & 31) of A and column (l & 31) of B, the 16 bytes of K-group (l >> 5) — the same
//     K positions on both sides — and D has column = lane & 31 (B index), row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5) (A index)
//  2. v_mfma_f32_32x32x16_f16 with only element 0 of lanes 0..31 non-zero on both sides is the rank-1 product a[m]*b[n], exact in f32
//  3. v_perm_b32 selector bytes 0x0C / 0x0D give 0x00 / 0xFF
//   hipcc --offload-arch=gfx950 -O2 -o scripts/probes/_bin/mmq_probe scripts/probes/mmq_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int   i32x4  __attribute__((ext_vector_type(4)));
typedef int   i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void k_probe(const int8_t * A /* [32][32] m,k */, const int8_t * B /* [32][32] n,k */, int * D /* [64][16] */,
                        const _Float16 * sa, const _Float16 * sb, float * SD, uint32_t * P) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    const i32x4 a = *(const i32x4 *) (A + r*32 + h*16);
    const i32x4 b = *(const i32x4 *) (B + r*32 + h*16);
    const i32x16 z = { 0 };
    const i32x16 d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
    for (int i = 0; i < 16; i++) D[l*16 + i] = d[i];
    h8 va = { 0 }, vb = { 0 };
    va[0] = h == 0 ? sa[r] : (_Float16) 0.0f;
    vb[0] = h == 0 ? sb[r] : (_Float16) 0.0f;
    const f32x16 zf = { 0 };
    const f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, zf, 0, 0, 0);
    for (int i = 0; i < 16; i++) SD[l*16 + i] = s[i];
    if (l < 16) {
        const uint32_t t = (((uint32_t) l * 0x00204081u) & 0x01010101u) | 0x0C0C0C0Cu;
        P[l] = __builtin_amdgcn_perm(0x12345678u, 0x9ABCDEF0u, t);
    }
}

int main() {
    int8_t hA[1024], hB[1024]; _Float16 ha[32], hb[32];
    srand(1);
    for (int i = 0; i < 1024; i++) { hA[i] = (int8_t) (rand() % 255 - 127); hB[i] = (int8_t) (rand() % 255 - 127); }
    for (int i = 0; i < 32; i++) { ha[i] = (_Float16) (0.001f * (i + 1) * (i % 3 ? 1 : -1)); hb[i] = (_Float16) (0.37f / (i + 1)); }
    hb[5] = (_Float16) 3e-6f;                                   // an f16 subnormal
    int8_t * dA, * dB; int * dD; _Float16 * da, * db; float * dS; uint32_t * dP;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096); hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dS, 4096); hipMalloc(&dP, 64);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, da, db, dS, dP);
    int hD[1024]; float hS[1024]; uint32_t hP[16];
    if (hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: kernel did not run\n"); return 1; }
    hipMemcpy(hS, dS, 4096, hipMemcpyDeviceToHost); hipMemcpy(hP, dP, 64, hipMemcpyDeviceToHost);
    int bad = 0, bad_t = 0, bad_s = 0, bad_p = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
        const int n = l & 31, m = (r & 3) + 8*(r >> 2) + 4*(l >> 5);
        int ref = 0, ref_t = 0;
        for (int k = 0; k < 32; k++) { ref += (int) hA[m*32 + k] * hB[n*32 + k]; ref_t += (int) hA[n*32 + k] * hB[m*32 + k]; }
        if (hD[l*16 + r] != ref) bad++;
        if (hD[l*16 + r] != ref_t) bad_t++;
        const float want = (float) ha[m] * (float) hb[n];
        if (hS[l*16 + r] != want) bad_s++;
    }
    for (int l = 0; l < 16; l++) {
        uint32_t want = 0;
        for (int b = 0; b < 4; b++) if (l & (1 << b)) want |= 0xFFu << (8*b);
        if (hP[l] != want) bad_p++;
    }
    printf("%s: i8 MFMA operand / result layout (mismatches %d of 1024; as the transpose %d)\n", bad == 0 ? "PASS" : "FAIL", bad, bad_t);
    printf("%s: rank-1 f16 scale MFMA exact (mismatches %d of 1024)\n", bad_s == 0 ? "PASS" : "FAIL", bad_s);
    printf("%s: v_perm_b32 0x0C / 0x0D selectors (mismatches %d of 16; e.g. sel(5) = %08x)\n", bad_p == 0 ? "PASS" : "FAIL", bad_p, hP[5]);
    return bad || bad_s || bad_p;
}
