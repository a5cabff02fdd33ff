// Probe for an ARITHMETIC replacement of the f16-table GELU (ggml_vec_gelu_f32, ggml-cpu/vec.h:987-1000; 25 % of the encoder's fc1 launch
// goes into its 64 gathers per lane, profiles/r03b_fc1_epilogue.txt).  The table has 65 536 entries — one per f16 input — so a formula
// can be checked against it exhaustively.  Candidates, all on u = sqrt(2/pi) x (1 + 0.044715 x^2) computed with the table's own
// operations (bit-identical u):
//   A  y = 0.5 x (1 + tanhf(u))            device libm tanhf
//   B  y = x / (1 + expf(-2u))             no cancellation for x < 0
//   C  y = x * rcp-free sigmoid via exp2: x / (1 + exp2(-2u log2 e)) with the hardware v_exp_f32
// For each: entries whose f16 rounding differs from the table, and how many of those a margin rule would have sent to the table
// ("unsafe": the f32 result lies within `margin` f32-ulps-of-y, scaled by |x / y| for A, of an f16 rounding boundary).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/probes/gelu_probe.hip -o /tmp/gelu_probe && /tmp/gelu_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float gelu_host(float x) { const float inner = fmaf(0.044715f*x, x, 1.0f); return 0.5f*x*(1.0f + tanhf(0.79788456080286535587989211986876f*x*inner)); }

__device__ __forceinline__ float u_of(float x) { const float inner = __builtin_fmaf(0.044715f*x, x, 1.0f); return 0.79788456080286535587989211986876f*x*inner; }

// distance (in units of half an f16 ulp of y, 0..1) of f32 y from the nearest f16 rounding boundary; normal f16 range only
__device__ __forceinline__ float boundary_dist(float y) {
    const _Float16 h = (_Float16) y;
    const float lo = (float) h;
    unsigned b; __builtin_memcpy(&b, &lo, 4);
    const unsigned eb = (b & 0x7F800000u) - (11u << 23);
    float halfulp; __builtin_memcpy(&halfulp, &eb, 4);
    return (halfulp - fabsf(y - lo)) / halfulp;          // 0 = on a boundary, 1 = exactly representable
}

__global__ void k_probe(const unsigned short * tab, unsigned * out /* [3][4] */, float margin) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    unsigned short hi = (unsigned short) i; _Float16 xh; __builtin_memcpy(&xh, &hi, 2);
    const float x = (float) xh;
    if (!(x > -10.0f && x < 10.0f)) return;              // the caller's clamps take the rest; NaN excluded
    const float u = u_of(x);
    float y[3];
    y[0] = 0.5f*x*(1.0f + tanhf(u));
    y[1] = x / (1.0f + expf(-2.0f*u));
    y[2] = x / (1.0f + __builtin_amdgcn_exp2f(-2.0f*u*1.4426950408889634f));
    for (int c = 0; c < 3; c++) {
        const _Float16 yh = (_Float16) y[c]; unsigned short got; __builtin_memcpy(&got, &yh, 2);
        const bool sub = fabsf(y[c]) < 6.103515625e-05f;                       // f16 subnormal range: always the table
        const float scale = c == 0 ? fmaxf(1.0f, fabsf(0.5f*x) / fmaxf(fabsf(y[c]), 1e-30f)) : 1.0f;
        const bool unsafe = sub || boundary_dist(y[c]) < margin * scale * (1.0f / 4096.0f);   // margin in f32 ulps of y against the 2^12 ulps of half an f16 ulp
        atomicAdd(&out[c*4 + 0], 1u);
        if (got != tab[i]) { atomicAdd(&out[c*4 + 1], 1u); if (!unsafe) atomicAdd(&out[c*4 + 2], 1u); }
        if (unsafe) atomicAdd(&out[c*4 + 3], 1u);
    }
}

int main() {
    unsigned short * tab = (unsigned short *) malloc(65536 * 2);
    for (int i = 0; i < 65536; i++) { unsigned short h = (unsigned short) i; _Float16 x; memcpy(&x, &h, 2); _Float16 y = (_Float16) gelu_host((float) x); memcpy(&tab[i], &y, 2); }
    unsigned short * d_tab; unsigned * d_out; CK(hipMalloc(&d_tab, 65536 * 2)); CK(hipMalloc(&d_out, 48));
    CK(hipMemcpy(d_tab, tab, 65536 * 2, hipMemcpyHostToDevice));
    const char * names[3] = { "A 0.5 x (1 + tanhf(u))", "B x / (1 + expf(-2u))", "C x / (1 + v_exp_f32(-2u log2e))" };
    for (float margin : { 4.0f, 8.0f, 16.0f, 32.0f }) {
        CK(hipMemset(d_out, 0, 48));
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), 0, 0, d_tab, d_out, margin);
        CK(hipDeviceSynchronize());
        unsigned o[12]; CK(hipMemcpy(o, d_out, 48, hipMemcpyDeviceToHost));
        for (int c = 0; c < 3; c++)
            printf("margin %4.0f ulp  %-34s inputs %5u  differ from the table %4u  of them NOT flagged unsafe %3u  flagged unsafe (would read the table) %5u = %.2f %%\n",
                   margin, names[c], o[c*4], o[c*4+1], o[c*4+2], o[c*4+3], 100.0 * o[c*4+3] / o[c*4]);
    }
    return 0;
}
