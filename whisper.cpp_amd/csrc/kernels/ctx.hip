// Context, launch emission, profiling, host-side re-layout of quantized blocks.
#include "common.h"
#include <hip/hip_ext.h>
#include <cxxabi.h>
#include <chrono>
#include <mutex>
#include <atomic>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void mi355x_set_error(const char * fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    if (getenv("GGML_MI355X_DEBUG")) fprintf(stderr, "mi355x: %s\n", g_err);
}
extern "C" const char * mi355x_last_error(void) { return g_err; }

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { mi355x_set_error("%s failed: %s", #call, hipGetErrorString(e_)); return (int) e_; } } while (0)

extern "C" int mi355x_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) != hipSuccess) continue;
        if (strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
        else return ok;   // devices are homogeneous on the target nodes; stop at the first foreign one
    }
    return ok;
}

// f32 GELU exactly as ggml_gelu_f32 (ggml-cpu/vec.h:968-970); evaluated on the HOST so that the table is
// produced by the same libm tanhf as the reference's ggml_table_gelu_f16 (ggml-cpu/ggml-cpu.c table init)
static inline float gelu_f32_host(float x) {
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    // the reference binary is built with gcc's default -ffp-contract=fast, which turns (A*x)*x + 1 into one fma;
    // with the explicit fmaf all 65536 table entries equal the reference's (tests/test_oracle.py, tests/test_host.py)
    const float inner = fmaf(GELU_COEF_A*x, x, 1.0f);
    return 0.5f*x*(1.0f + tanhf(SQRT_2_OVER_PI*x*inner));
}

extern "C" void mi355x_gelu_table_host(uint16_t * out) {
    for (int i = 0; i < 65536; i++) {
        uint16_t h = (uint16_t) i; _Float16 x; memcpy(&x, &h, 2);
        _Float16 y = (_Float16) gelu_f32_host((float) x);
        memcpy(&out[i], &y, 2);
    }
}

extern "C" mi355x_ctx * mi355x_ctx_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { mi355x_set_error("no HIP device %d", device); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    mi355x_ctx * ctx = new mi355x_ctx();
    ctx->device = device;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) == hipSuccess) ctx->n_cu = p.multiProcessorCount;
    // (one XCD-masked stream per state — hipExtStreamCreateWithCUMask — was measured in round 3 and lost: HISTORY.md)
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
    std::vector<uint16_t> tab(65536);
    mi355x_gelu_table_host(tab.data());
    if (hipMalloc((void **) &ctx->gelu_tab, 65536*2) != hipSuccess ||
        hipMemcpy(ctx->gelu_tab, tab.data(), 65536*2, hipMemcpyHostToDevice) != hipSuccess) {
        mi355x_set_error("gelu table upload failed"); (void) hipStreamDestroy(ctx->stream); delete ctx; return nullptr;
    }
    return ctx;
}

extern "C" void mi355x_ctx_destroy(mi355x_ctx * ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) mi355x_flush_pending(ctx);
    (void) hipStreamSynchronize(ctx->stream);
    for (auto & e : ctx->ev_pool) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
    if (ctx->scratch)  (void) hipFree(ctx->scratch);
    for (void * r : ctx->scratch_retired) (void) hipFree(r);
    if (ctx->gelu_tab) (void) hipFree(ctx->gelu_tab);
    if (ctx->mel_tab)  (void) hipFree(ctx->mel_tab);
    if (ctx->qact)     (void) hipFree(ctx->qact);
    (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" void * mi355x_ctx_stream(mi355x_ctx * ctx) { return (void *) ctx->stream; }

// A wide do-nothing launch.  The first chip-wide dispatch after the GPU has sat idle for a few hundred microseconds (the
// host-side part of a decode step) stalls ~30 us (profiles: gap before the first mat-vec of every step, while the
// 1-workgroup kernels in front of it start immediately).  The backend issues this on its upload stream when the first input
// of a step arrives, i.e. while the host is still busy, so the stall is taken off the critical path.
__global__ void __launch_bounds__(64) k_wake(int * sink) { if (sink && threadIdx.x == 1024) *sink = 0; }
extern "C" int mi355x_wake(void * stream, int nblocks) {
    k_wake<<<dim3(nblocks > 0 ? nblocks : 1024), dim3(64), 0, (hipStream_t) stream>>>(nullptr);
    return (int) hipGetLastError();
}

// kernel-anatomy stamps (debug): allocated on first use when GGML_MI355X_KTIME=1, else kernels get a null pointer
void * mi355x_debug_stamps(mi355x_ctx * ctx) {
    static const bool on = getenv("GGML_MI355X_KTIME") && atoi(getenv("GGML_MI355X_KTIME"));
    if (!on) return nullptr;
    if (!ctx->dbg_stamps && hipMalloc(&ctx->dbg_stamps, 16*8) == hipSuccess) (void) hipMemsetAsync(ctx->dbg_stamps, 0, 16*8, ctx->stream);
    return ctx->dbg_stamps;
}
extern "C" int mi355x_debug_read_stamps(mi355x_ctx * ctx, unsigned long long * out16) {
    if (!ctx->dbg_stamps) return MI355X_E_UNSUPPORTED;
    (void) hipStreamSynchronize(ctx->stream);
    return (int) hipMemcpy(out16, ctx->dbg_stamps, 16*8, hipMemcpyDeviceToHost);
}

static void prof_drain(mi355x_ctx * ctx) {
    for (auto & p : ctx->ev_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[p.ev].first, ctx->ev_pool[p.ev].second) == hipSuccess) {
            prof_acc & a = ctx->prof_rows[p.name];
            a.calls++; a.ms += ms; a.bytes += p.bytes; a.flops += p.flops;
        }
    }
    ctx->ev_pending.clear();
}

// ---- test options (mi355x_kernels.h: mi355x_test_option): kernel-shape variants a test compares with the default inside one process ----
static std::atomic<int> g_test_opt[MI355X_OPT_COUNT];
static std::atomic<bool> g_test_opt_set[MI355X_OPT_COUNT];
extern "C" void mi355x_test_option(int opt, int value, int set) {
    if (opt < 0 || opt >= MI355X_OPT_COUNT) return;
    g_test_opt[opt].store(value); g_test_opt_set[opt].store(set != 0);
}
int mi355x_opt(int opt, int def) { return g_test_opt_set[opt].load(std::memory_order_relaxed) ? g_test_opt[opt].load(std::memory_order_relaxed) : def; }

extern "C" int mi355x_flush(mi355x_ctx * ctx) { return mi355x_flush_pending(ctx); }
extern "C" int mi355x_last_launch_mirrored(mi355x_ctx * ctx) { const int r = ctx->last_mirrored; ctx->last_mirrored = 0; return r; }


extern "C" int mi355x_ctx_synchronize(mi355x_ctx * ctx) {
    (void) hipSetDevice(ctx->device);
    { const int rc = mi355x_flush_pending(ctx); if (rc) return rc; }
    HIP_OK(hipStreamSynchronize(ctx->stream));
    if (ctx->prof) prof_drain(ctx);
    return 0;
}

// ---- scratch -----------------------------------------------------------------------------------
void mi355x_scratch_reset(mi355x_ctx * ctx) { ctx->scratch_used = 0; }

void * mi355x_scratch_alloc(mi355x_ctx * ctx, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t) 255;
    if (ctx->scratch_used + bytes > ctx->scratch_size) {
        // grow: allocate a new arena; the old one may still be in use by enqueued kernels, so drain first.
        // (happens a handful of times during warm-up, never in steady state)
        size_t want = ctx->scratch_used + bytes;
        size_t nsz = ctx->scratch_size ? ctx->scratch_size : ((size_t) 64 << 20);
        while (nsz < want) nsz *= 2;
        (void) mi355x_flush_pending(ctx);                // a held-back GEMM reads rows in the arena that is about to be retired: it leaves first
        (void) hipStreamSynchronize(ctx->stream);
        void * nptr = nullptr;
        if (hipMalloc(&nptr, nsz) != hipSuccess) { mi355x_set_error("scratch alloc of %zu bytes failed", nsz); return nullptr; }
        // arenas retired by EARLIER growths are idle now (the stream was just drained); the current one may hold pointers handed
        // out earlier in this op group (e.g. part_o before part_ml), so it stays alive until the next growth / context destroy
        for (void * r : ctx->scratch_retired) (void) hipFree(r);
        ctx->scratch_retired.clear();
        if (ctx->scratch) { if (ctx->scratch_used > 0) ctx->scratch_retired.push_back(ctx->scratch); else (void) hipFree(ctx->scratch); }
        ctx->scratch = nptr; ctx->scratch_size = nsz; ctx->scratch_used = 0;
    }
    void * p = (char *) ctx->scratch + ctx->scratch_used;
    ctx->scratch_used += bytes;
    return p;
}

// ---- emission ----------------------------------------------------------------------------------
int mi355x_emit(mi355x_ctx * ctx, const char * name, const void * func, dim3 grid, dim3 block, uint32_t shmem,
                const void * args, uint32_t arg_size, double algo_bytes, double algo_flops) {
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return 0;
    { const int rc = mi355x_flush_pending(ctx); if (rc) return rc; }      // held-back launches go first: stream order is program order
    int ev = -1;
    void * kargs[1] = { (void *) args };
    hipError_t e;
    ctx->n_eager++;
    if (ctx->prof) {
        // hipExtLaunchKernel attaches the two events to the dispatch itself: they carry the kernel's own begin / end
        // timestamps (what rocprofv3 --kernel-trace reports), not the time of separately enqueued event markers,
        // which adds ~5 us to a 3 us kernel
        ev = (int) ctx->ev_pending.size();
        if (ev >= (int) ctx->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
            ctx->ev_pool.push_back({a, b});
        }
        e = hipExtLaunchKernel(func, grid, block, kargs, shmem, ctx->stream, ctx->ev_pool[ev].first, ctx->ev_pool[ev].second, 0);
    } else {
        e = hipLaunchKernel(func, grid, block, kargs, shmem, ctx->stream);
    }
    if (e != hipSuccess) { mi355x_set_error("launch of %s failed: %s", name, hipGetErrorString(e)); return (int) e; }
    if (ctx->prof) {
        // profile rows are keyed by the kernel's own (demangled) symbol, i.e. exactly the name rocprofv3 reports
        static std::map<const void *, const char *> names;
        static std::mutex names_mtx;
        {
            std::lock_guard<std::mutex> lk(names_mtx);
            auto it = names.find(func);
            if (it == names.end()) {
                const char * mangled = hipKernelNameRefByPtr(func, ctx->stream);
                const char * shown = name;
                if (mangled) {
                    int status = 0;
                    char * dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
                    shown = strdup(status == 0 && dem ? dem : mangled);       // lives for the process lifetime
                    free(dem);
                }
                it = names.emplace(func, shown).first;
            }
            name = it->second;
        }
        ctx->ev_pending.push_back({name, ev, algo_bytes, algo_flops});
        if (ctx->ev_pending.size() >= 4096) { HIP_OK(hipStreamSynchronize(ctx->stream)); prof_drain(ctx); }
    }
    return 0;
}

extern "C" uint64_t mi355x_eager_count(mi355x_ctx * ctx) { return ctx->n_eager; }

extern "C" void mi355x_prof_enable(mi355x_ctx * ctx, int on) { ctx->prof = on != 0; }
extern "C" void mi355x_prof_reset(mi355x_ctx * ctx) { (void) mi355x_flush_pending(ctx); (void) hipStreamSynchronize(ctx->stream); prof_drain(ctx); ctx->prof_rows.clear(); }
extern "C" int  mi355x_prof_report(mi355x_ctx * ctx, mi355x_prof_row * rows, int cap) {
    (void) mi355x_flush_pending(ctx);
    (void) hipStreamSynchronize(ctx->stream);
    prof_drain(ctx);
    int n = 0;
    for (auto & kv : ctx->prof_rows) {
        if (n >= cap) break;
        rows[n].name = kv.first.c_str(); rows[n].calls = kv.second.calls; rows[n].total_ms = kv.second.ms;
        rows[n].algo_bytes = kv.second.bytes; rows[n].algo_flops = kv.second.flops;
        n++;
    }
    return n;
}

extern "C" int mi355x_memset(mi355x_ctx * ctx, void * dptr, int value, size_t n) {
    (void) hipSetDevice(ctx->device);
    { const int rc = mi355x_flush_pending(ctx); if (rc) return rc; }
    HIP_OK(hipMemsetAsync(dptr, value, n, ctx->stream));
    return 0;
}

// ---- host re-layout ------------------------------------------------------------------------------
extern "C" int mi355x_type_is_quantized(int type) {
    return type == MI355X_TYPE_Q4_0 || type == MI355X_TYPE_Q5_0 || type == MI355X_TYPE_Q8_0 || type == MI355X_TYPE_Q4_K;
}
extern "C" size_t mi355x_type_row_bytes(int type, int64_t ne0) {
    const int bs = type_block(type), ts = type_size(type);
    if (!ts || ne0 % bs) return 0;
    return (size_t) (ne0 / bs) * ts;
}

// ggml block structs (ggml/src/ggml-common.h): q4_0 {f16 d; u8 qs[16]}, q5_0 {f16 d; u8 qh[4]; u8 qs[16]},
// q8_0 {f16 d; i8 qs[32]}, q4_K {f16 d; f16 dmin; u8 scales[12]; u8 qs[128]}
template <bool TO_PLANAR>
static int repack(int type, const uint8_t * a, uint8_t * b, int64_t nelements) {
    const int bs = type_block(type);
    if (bs == 1 || nelements % bs) return MI355X_E_UNSUPPORTED;
    const int64_t nb = nelements / bs;
    // `blk` points into the ggml-struct side, plane pointers into the planar side
    const uint8_t * src = a; uint8_t * dst = b;
    auto cp = [&](int64_t blk_off, int64_t plane_off, size_t n) {
        if (TO_PLANAR) memcpy(dst + plane_off, src + blk_off, n);
        else           memcpy(dst + blk_off, src + plane_off, n);
    };
    switch (type) {
        case MI355X_TYPE_Q4_0:
            for (int64_t i = 0; i < nb; i++) { cp(i*18 + 2, i*16, 16); cp(i*18, nb*16 + i*2, 2); }
            return 0;
        case MI355X_TYPE_Q5_0:
            for (int64_t i = 0; i < nb; i++) { cp(i*22 + 6, i*16, 16); cp(i*22 + 2, nb*16 + i*4, 4); cp(i*22, nb*20 + i*2, 2); }
            return 0;
        case MI355X_TYPE_Q8_0:
            for (int64_t i = 0; i < nb; i++) { cp(i*34 + 2, i*32, 32); cp(i*34, nb*32 + i*2, 2); }
            return 0;
        case MI355X_TYPE_Q4_K:
            for (int64_t i = 0; i < nb; i++) { cp(i*144 + 16, i*128, 128); cp(i*144 + 4, nb*128 + i*12, 12); cp(i*144, nb*140 + i*4, 4); }
            return 0;
        default: return MI355X_E_UNSUPPORTED;
    }
}
extern "C" int mi355x_repack_to_planar(int type, const void * ggml_blocks, void * planar, int64_t nelements) {
    return repack<true>(type, (const uint8_t *) ggml_blocks, (uint8_t *) planar, nelements);
}
extern "C" int mi355x_repack_from_planar(int type, const void * planar, void * ggml_blocks, int64_t nelements) {
    return repack<false>(type, (const uint8_t *) planar, (uint8_t *) ggml_blocks, nelements);
}

// ---- buffer checksum (multi-GPU weight distribution: every replica's weights must equal rank 0's) ------------------------
// out[0] = sum of the 64-bit words, out[1] = sum of word * (2*index + 1), both mod 2^64: independent of the summation order,
// sensitive to any changed, missing or transposed word.  Tail bytes (< 8) are folded in as one zero-padded word.
__global__ void __launch_bounds__(256) k_checksum(const uint64_t * __restrict__ p, size_t nwords, const uint8_t * tail, int ntail, unsigned long long * out) {
    unsigned long long s0 = 0, s1 = 0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t) gridDim.x * 256) {
        const unsigned long long w = p[i];
        s0 += w; s1 += w * (2ull * i + 1ull);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && ntail > 0) {
        unsigned long long w = 0;
        for (int b = 0; b < ntail; b++) w |= (unsigned long long) tail[b] << (8*b);
        s0 += w; s1 += w * (2ull * nwords + 1ull);
    }
    // wave reduction (integer adds: order does not matter), one atomic pair per wave
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], s0); atomicAdd(&out[1], s1); }
}
extern "C" int mi355x_checksum(void * stream, const void * dptr, size_t nbytes, void * dev_out16) {
    HIP_OK(hipMemsetAsync(dev_out16, 0, 16, (hipStream_t) stream));
    if (nbytes == 0) return 0;
    const size_t nwords = nbytes / 8;
    const int ntail = (int) (nbytes % 8);
    size_t nb = (nwords + 255) / 256; if (nb < 1) nb = 1; if (nb > 4096) nb = 4096;
    k_checksum<<<dim3((uint32_t) nb), dim3(256), 0, (hipStream_t) stream>>>((const uint64_t *) dptr, nwords, (const uint8_t *) dptr + nwords*8, ntail, (unsigned long long *) dev_out16);
    return (int) hipGetLastError();
}

// ---- batched small uploads ------------------------------------------------------------------------------------------
// The scheduler writes the graph inputs of a decode step (token ids, positions, mask: 4 B .. a few KB each) into device tensors
// right before graph_compute.  As three H2D copies they are three serialized blit kernels plus a cross-stream event in front of
// the step's first kernel; here the host only memcpy()s them into pinned, device-mapped memory and ONE launch on the compute
// stream moves all of them (each workgroup reads its record over PCIe and writes the device tensor).
struct ScatterArgs { void * dst[MI355X_SCATTER_MAX]; const void * src[MI355X_SCATTER_MAX]; uint32_t size[MI355X_SCATTER_MAX]; int n; };
__global__ void __launch_bounds__(256) k_scatter_upload(const ScatterArgs a) {
    const int r = blockIdx.x;
    if (r >= a.n) return;
    char * d = (char *) a.dst[r]; const char * s = (const char *) a.src[r];
    const uint32_t n = a.size[r];
    if ((((uintptr_t) d | (uintptr_t) s) & 15) == 0) {
        const uint32_t n16 = n >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += 256) ((uint4 *) d)[i] = ((const uint4 *) s)[i];
        for (uint32_t i = (n16 << 4) + threadIdx.x; i < n; i += 256) d[i] = s[i];
    } else if ((((uintptr_t) d | (uintptr_t) s | n) & 3) == 0) {
        for (uint32_t i = threadIdx.x; i < (n >> 2); i += 256) ((uint32_t *) d)[i] = ((const uint32_t *) s)[i];
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += 256) d[i] = s[i];
    }
}
extern "C" int mi355x_scatter_upload(void * stream, int n, void * const * dst, const void * const * src_dev, const uint32_t * sizes) {
    for (int i0 = 0; i0 < n; i0 += MI355X_SCATTER_MAX) {
        ScatterArgs a; memset(&a, 0, sizeof(a));
        a.n = n - i0 < MI355X_SCATTER_MAX ? n - i0 : MI355X_SCATTER_MAX;
        for (int i = 0; i < a.n; i++) { a.dst[i] = dst[i0 + i]; a.src[i] = src_dev[i0 + i]; a.size[i] = sizes[i0 + i]; }
        k_scatter_upload<<<dim3((uint32_t) a.n), dim3(256), 0, (hipStream_t) stream>>>(a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { mi355x_set_error("scatter upload launch failed: %s", hipGetErrorString(e)); return (int) e; }
    }
    return 0;
}
