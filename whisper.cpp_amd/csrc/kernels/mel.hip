// Log-mel front end on the GPU: whisper's log_mel_spectrogram (src/whisper.cpp:3046-3283) for 16 kHz PCM that
// is already resident in HBM — Hann window (400), hop 160, 201-bin power spectrum, mel filterbank, log10, clamp to (max - 8), (x + 4) / 4.
//
// The reference runs a recursive radix-2 FFT that ends in 25-point DFTs on 4 host threads (9.7 ms for the 11 s of samples/jfk.wav,
// 16 ms per 30 s chunk).  At 1.5 ms per encode that would be the per-chunk bottleneck.  Here a frame is one workgroup and a bin is one
// thread evaluating the 400-term DFT directly from the same sin / cos table (table index (k n) mod 400, exactly the reference's dft()):
// 201 x 400 x 2 FMAs per frame is nothing for the chip, and the direct sum is at least as accurate as the recursion (measured against
// the reference on jfk.wav: max |diff| 1.7e-5 in normalised mel units).  Accumulation widths follow the reference: f32 spectrum, f64
// filterbank sum and log10.
#include "common.h"
#include <math.h>

#define MEL_N_FFT 400
#define MEL_HOP   160
#define MEL_BINS  201

struct MelArgs {
    const float * pcm; int n_samples;           // un-padded samples
    const float * filt; int n_mel;              // [n_mel][201]
    const float * sincos;                       // [2][400]: sin(2 pi i / 400), cos(...), computed on the host with sinf / cosf like the reference
    const float * hann;                         // [400]
    float * out; int n_len; int n_live;         // out[j * n_len + i]; frames >= n_live hold log10(1e-10)
    unsigned int * maxbits;                     // running maximum of the log-mel values (monotone float -> uint mapping)
};

__device__ __forceinline__ unsigned int f2ord(float f) { const unsigned int u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned int o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ void __launch_bounds__(256) k_log_mel(const MelArgs a) {
    __shared__ float x[MEL_N_FFT], cs[MEL_N_FFT], sn[MEL_N_FFT], pw[MEL_BINS + 3];
    const int i = blockIdx.x, tid = threadIdx.x;
    if (i >= a.n_live) {
        // no sample reaches this frame: log10(max(0, 1e-10)) = -10 (src/whisper.cpp:3173-3179)
        for (int j = tid; j < a.n_mel; j += 256) a.out[(int64_t) j * a.n_len + i] = -10.0f;
        if (tid == 0) atomicMax(a.maxbits, f2ord(-10.0f));
        return;
    }
    // frame i of the padded signal: 200 reflected samples in front (pad[200 - k] = pcm[k]), zeros behind the audio
    for (int j = tid; j < MEL_N_FFT; j += 256) {
        const int64_t p = (int64_t) i * MEL_HOP + j - MEL_N_FFT / 2;
        float v = 0.0f;
        if (p < 0) { const int64_t k = -p; v = k < a.n_samples ? a.pcm[k] : 0.0f; }
        else if (p < a.n_samples) v = a.pcm[p];
        x[j] = a.hann[j] * v;
        sn[j] = a.sincos[j]; cs[j] = a.sincos[MEL_N_FFT + j];
    }
    __syncthreads();
    if (tid < MEL_BINS) {
        float re = 0.0f, im = 0.0f;
        int idx = 0;                                   // (k * n) mod 400, advanced incrementally
        for (int n = 0; n < MEL_N_FFT; n++) {
            re += x[n] * cs[idx];
            im -= x[n] * sn[idx];
            idx += tid; if (idx >= MEL_N_FFT) idx -= MEL_N_FFT;
        }
        pw[tid] = re*re + im*im;
    }
    __syncthreads();
    for (int j = tid; j < a.n_mel; j += 256) {
        const float * f = a.filt + (int64_t) j * MEL_BINS;
        double sum = 0.0;
        int k = 0;
        for (; k < MEL_BINS - 3; k += 4) sum += (double) (pw[k]*f[k] + pw[k + 1]*f[k + 1] + pw[k + 2]*f[k + 2] + pw[k + 3]*f[k + 3]);
        for (; k < MEL_BINS; k++) sum += (double) (pw[k]*f[k]);
        const float v = (float) log10(fmax(sum, 1e-10));
        a.out[(int64_t) j * a.n_len + i] = v;
        atomicMax(a.maxbits, f2ord(v));
    }
}

struct MelNormArgs { float * out; int64_t n; const unsigned int * maxbits; };
__global__ void __launch_bounds__(256) k_log_mel_norm(const MelNormArgs a) {
    const double mmax = (double) ord2f(*a.maxbits) - 8.0;
    for (int64_t idx = (int64_t) blockIdx.x * 256 + threadIdx.x; idx < a.n; idx += (int64_t) gridDim.x * 256) {
        double v = (double) a.out[idx];
        if (v < mmax) v = mmax;
        a.out[idx] = (float) ((v + 4.0) / 4.0);
    }
}

extern "C" int mi355x_log_mel_n_len(int n_samples) { return (int) (((int64_t) n_samples + 16000 * 30) / MEL_HOP); }       // src/whisper.cpp:3232-3236

extern "C" int mi355x_log_mel(mi355x_ctx * ctx, const float * pcm_dev, int n_samples, const float * filters_dev, int n_mel, int n_fft_bins,
                              float * mel_dev, int n_len) {
    if (n_fft_bins != MEL_BINS || n_mel < 1 || n_mel > 1024 || n_samples < 1 || n_len != mi355x_log_mel_n_len(n_samples)) return MI355X_E_UNSUPPORTED;
    (void) hipSetDevice(ctx->device);
    if (!ctx->mel_tab) {
        // sin_vals / cos_vals / hann_window of whisper_global_cache (src/whisper.cpp:3005-3040), same libm calls
        std::vector<float> t(3 * MEL_N_FFT + 4);
        for (int i = 0; i < MEL_N_FFT; i++) {
            const double theta = (2 * M_PI * i) / MEL_N_FFT;
            t[i] = sinf(theta); t[MEL_N_FFT + i] = cosf(theta);
            t[2*MEL_N_FFT + i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / MEL_N_FFT));
        }
        if (hipMalloc((void **) &ctx->mel_tab, t.size() * 4) != hipSuccess) return (int) hipErrorOutOfMemory;
        if (hipMemcpy(ctx->mel_tab, t.data(), t.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return (int) hipErrorUnknown;
    }
    MelArgs a;
    a.pcm = pcm_dev; a.n_samples = n_samples; a.filt = filters_dev; a.n_mel = n_mel;
    a.sincos = ctx->mel_tab; a.hann = ctx->mel_tab + 2*MEL_N_FFT;
    a.out = mel_dev; a.n_len = n_len;
    const int64_t live = ((int64_t) n_samples + MEL_N_FFT / 2) / MEL_HOP + 1;            // log_mel_spectrogram_worker_thread: i < min(n_samples / frame_step + 1, n_len)
    a.n_live = (int) (live < n_len ? live : n_len);
    a.maxbits = (unsigned int *) (ctx->mel_tab + 3*MEL_N_FFT);
    { const int rc = mi355x_flush_pending(ctx); if (rc) return rc; }
    HIP_CHECK_RET(hipMemsetAsync(a.maxbits, 0, 4, ctx->stream));
    int rc = emit(ctx, "log_mel", k_log_mel, dim3((uint32_t) n_len), dim3(256), 0, a, (double) n_samples * 4 + (double) n_mel * n_len * 4, 2.0 * 2 * MEL_BINS * MEL_N_FFT * a.n_live);
    if (rc) return rc;
    const int64_t n = (int64_t) n_mel * n_len;
    const MelNormArgs na = { mel_dev, n, a.maxbits };
    return emit(ctx, "log_mel_norm", k_log_mel_norm, dim3((uint32_t) ((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, na, (double) n * 8, 0);
}
