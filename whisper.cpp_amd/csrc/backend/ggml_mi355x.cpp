// ggml backend plugin for AMD Instinct MI355X (gfx950).  Host side only: registration, buffers, graph walk +
// fusion planner.  All device work goes through the C ABI of libmi355x_kernels.so
// (include/mi355x_kernels.h).  Boundary documentation: include/ggml_mi355x.h.
//
// Reference interface being implemented: ggml/src/ggml-backend-impl.h (vtables), loader
// ggml/src/ggml-backend-reg.cpp:220-264, scheduler call sites ggml/src/ggml-backend.cpp:1594-1780.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"

#include <hip/hip_runtime_api.h>

#include "ggml_mi355x.h"
#include "mi355x_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#define MI_MAX_DEVICES 16
#define MI_ALIGNMENT   256      // tensor alignment inside buffers (hipMalloc itself is 256-B aligned or better)

static bool env_flag(const char * name, bool def) {
    const char * v = getenv(name);
    if (!v || !*v) return def;
    return !(v[0] == '0' || v[0] == 'n' || v[0] == 'N' || v[0] == 'f' || v[0] == 'F');
}
static bool g_debug() { static bool d = env_flag("GGML_MI355X_DEBUG", false); return d; }
#define MI_LOG(...) do { if (g_debug()) { fprintf(stderr, "ggml-mi355x: " __VA_ARGS__); fputc('\n', stderr); } } while (0)

// ---------------------------------------------------------------------------------------------------
// device / registry state
// ---------------------------------------------------------------------------------------------------
struct mi_device_ctx {
    int         index;
    std::string name;          // "MI355X0"
    std::string description;   // from hipDeviceProp
    ggml_backend_buffer_type buft;
};

struct mi_buffer_ctx {
    int    device;
    void * base;
    size_t size;
};

struct mi_weight_rec { int device; void * base; size_t size; ggml_backend_buffer_t buf; };
static std::mutex                 g_weights_mtx;
static std::vector<mi_weight_rec> g_buffers;       // every live device buffer, in allocation order

static ggml_backend_reg           g_reg;
static ggml_backend_device        g_devices[MI_MAX_DEVICES];
static mi_device_ctx              g_device_ctx[MI_MAX_DEVICES];
static int                        g_n_devices = -1;

static bool is_quant_type(ggml_type t) { return mi355x_type_is_quantized((int) t) != 0; }
// ggml_backend_mi355x_defer_weights: weight uploads issued BY THE CALLING THREAD are skipped (they arrive by broadcast).  Scoped to
// the thread that creates the replica's context — whisper_init_* runs its set_tensor loop on the caller's thread (src/whisper.cpp:1934-
// 1938) — so another context loading concurrently on another thread, or on another device, is never affected.
static thread_local int       t_defer_weights = 0;
static std::atomic<uint64_t>  g_deferred_bytes{0};    // bytes skipped so far (visible through GGML_MI355X_DEBUG and ggml_backend_mi355x_deferred_bytes)

// f16 copies of quantized WEIGHTS tensors that meet wide activations (encoder, cross-attention K/V, prompt): made once by
// mi355x_dequant_f16 the first time such a tensor reaches the MFMA path, kept until its buffer is written or freed.
// 2 bytes/weight of HBM buys a GEMM inner loop without dequantization (GGML_MI355X_F16_SHADOW_MB caps the total, 0 = off).
struct mi_shadow { void * f16; size_t bytes; const void * buf_base; int device; int type; int64_t ne0, ne1; };
static std::mutex                                   g_shadow_mtx;
static std::unordered_map<const void *, mi_shadow>  g_shadows;
static std::atomic<size_t>                          g_shadow_count{0};
static size_t                                       g_shadow_bytes = 0;

// forget (and free) the copies that belong to one buffer (buf_base) or one device (buf_base == nullptr)
static void mi_shadows_drop(int device, const void * buf_base) {
    if (g_shadow_count.load() == 0) return;
    std::lock_guard<std::mutex> lk(g_shadow_mtx);
    bool synced = false;
    for (auto it = g_shadows.begin(); it != g_shadows.end(); ) {
        if (it->second.device == device && (!buf_base || it->second.buf_base == buf_base)) {
            if (!synced) { (void) hipSetDevice(device); (void) hipDeviceSynchronize(); synced = true; }
            (void) hipFree(it->second.f16);
            g_shadow_bytes -= it->second.bytes;
            it = g_shadows.erase(it);
        } else ++it;
    }
    g_shadow_count.store(g_shadows.size());
}

// host-side time spent in the buffer callbacks (set/get/cpy: the per-step H2D of ids / positions / mask and the D2H of
// the logits row) and in synchronize — reported by ggml_backend_mi355x_host_times
static std::atomic<uint64_t> g_io_ns[4] = {};      // set_tensor, get_tensor, cpy_tensor, synchronize
static std::atomic<uint64_t> g_io_calls[4] = {};
struct io_timer {
    int slot; std::chrono::steady_clock::time_point t0;
    explicit io_timer(int s) : slot(s), t0(std::chrono::steady_clock::now()) {}
    ~io_timer() { g_io_ns[slot] += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_io_calls[slot]++; }
};

// GGML_MI355X_TRACE=1: where the host-side time of a step goes INSIDE the plugin (tests/native/step_trace.cpp reads it through
// ggml_backend_mi355x_trace and sets it beside the reference-side timeline it measures by interposing the scheduler's entry points).
// Slots (ns, calls): 0 supports_op, 1 supports_buft, 2 graph_compute entry -> first kernel launched, 3 graph_compute entry -> return,
// 4 graph_compute entry -> synchronize return (one step's whole device phase as the host sees it), 5 get_proc_address
static const bool g_trace = env_flag("GGML_MI355X_TRACE", false);
static std::atomic<uint64_t> g_trace_ns[8] = {};
static std::atomic<uint64_t> g_trace_calls[8] = {};
static inline uint64_t trace_now() { return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct trace_scope {
    int slot; uint64_t t0;
    explicit trace_scope(int s) : slot(s), t0(g_trace ? trace_now() : 0) {}
    ~trace_scope() { if (g_trace) { g_trace_ns[slot] += trace_now() - t0; g_trace_calls[slot]++; } }
};

// ---------------------------------------------------------------------------------------------------
// small host <-> device transfers.  The scheduler copies the graph inputs (token ids, positions, mask) into our buffers
// before EVERY decode step (ggml-backend.cpp:1625-1632) and whisper reads one logits row back after it (W:2957-2963).
// A synchronous hipMemcpy from pageable memory costs ~50 us per call; these go through a pinned ring buffer and an
// upload stream instead: set_tensor returns as soon as the copy is enqueued (the source has been copied into the ring, so
// the caller may reuse it), compute streams wait on the upload event, every other reader drains the upload stream first.
// ---------------------------------------------------------------------------------------------------
struct mi_io_rec { void * dst; uint32_t off, size; };
struct mi_io_ctx {
    std::mutex  mtx;
    hipStream_t stream = nullptr;
    hipEvent_t  ev = nullptr;              // last async COPY on `stream` (uploads too large for the deferred path)
    hipEvent_t  ev_flush = nullptr;        // last scatter launch of deferred uploads (on whichever stream flushed them)
    hipStream_t flush_stream = nullptr;    // the stream ev_flush was recorded on
    char *      pinned = nullptr;
    char *      pinned_dev = nullptr;      // the same memory as the device sees it
    size_t      cap = 0, off = 0;
    std::vector<mi_io_rec> pending;        // deferred small uploads: bytes are in the ring, one scatter launch moves them
    std::atomic<uint64_t> seq{0};          // number of uploads accepted so far (both paths)
    std::atomic<uint64_t> copy_seq{0};     // ... of which went through async copies on `stream`
    std::atomic<uint64_t> drained{0};      // uploads known to be complete
    uint64_t    wake_seq = 0;              // value of seq when the last graph_compute picked the uploads up
    uint64_t    flush_count = 0;           // scatter flushes so far
    bool        ok = false, tried = false, flushed_since_drain = false;
};
static mi_io_ctx g_io[MI_MAX_DEVICES];
#define MI_IO_SMALL (256u << 10)
#define MI_IO_DEFER (32u << 10)            // uploads up to this size wait in the ring for the next flush (graph inputs of a decode step)

static mi_io_ctx * mi_io(int device) {          // caller holds no lock; device already current
    mi_io_ctx & io = g_io[device];
    std::lock_guard<std::mutex> lk(io.mtx);
    if (!io.tried) {
        io.tried = true;
        io.cap = (size_t) 4 << 20;
        if (hipStreamCreateWithFlags(&io.stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&io.ev, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&io.ev_flush, hipEventDisableTiming) == hipSuccess &&
            hipHostMalloc((void **) &io.pinned, io.cap, hipHostMallocDefault) == hipSuccess) {
            io.ok = true;
            void * dp = nullptr;
            if (hipHostGetDevicePointer(&dp, io.pinned, 0) == hipSuccess) io.pinned_dev = (char *) dp;
            else (void) hipGetLastError();
        }
    }
    return io.ok ? &io : nullptr;
}
// one scatter launch on `stream` for everything that waits in the ring (io.mtx held)
static void mi_io_flush_locked(mi_io_ctx & io, hipStream_t stream) {
    if (io.pending.empty()) return;
    // flushes form a chain: ev_flush only remembers the LAST one, so a flush on another stream is ordered behind its predecessor
    // and whoever waits for the last one has waited for all of them (several whisper_states on one device)
    if (io.flush_stream && io.flush_stream != stream) (void) hipStreamWaitEvent(stream, io.ev_flush, 0);
    void * dst[64]; const void * src[64]; uint32_t sz[64];
    size_t i = 0;
    while (i < io.pending.size()) {
        int n = 0;
        for (; n < 64 && i < io.pending.size(); n++, i++) { dst[n] = io.pending[i].dst; src[n] = io.pinned_dev + io.pending[i].off; sz[n] = io.pending[i].size; }
        if (mi355x_scatter_upload((void *) stream, n, dst, src, sz) != 0) GGML_ABORT("ggml-mi355x: upload of graph inputs failed: %s", mi355x_last_error());
    }
    io.pending.clear();
    (void) hipEventRecord(io.ev_flush, stream);
    io.flush_stream = stream; io.flushed_since_drain = true; io.flush_count++;
}
// wait until every accepted upload has landed (readers other than the compute streams)
static void mi_io_drain(int device) {
    mi_io_ctx & io = g_io[device];
    if (!io.ok || io.drained.load() == io.seq.load()) return;
    std::lock_guard<std::mutex> lk(io.mtx);
    const uint64_t s = io.seq.load();
    mi_io_flush_locked(io, io.stream);
    (void) hipStreamSynchronize(io.stream);
    if (io.flushed_since_drain) { (void) hipEventSynchronize(io.ev_flush); io.flushed_since_drain = false; }
    io.drained.store(s); io.off = 0;
}
static bool mi_io_upload(int device, void * dst, const void * src, size_t size) {
    mi_io_ctx * io = mi_io(device);
    if (!io) return false;
    std::lock_guard<std::mutex> lk(io->mtx);
    const size_t need = (size + 255) & ~(size_t) 255;
    if (io->off + need > io->cap / 2) {            // ring full: everything that still reads from it must finish first
        mi_io_flush_locked(*io, io->stream);
        (void) hipStreamSynchronize(io->stream);
        if (io->flushed_since_drain) { (void) hipEventSynchronize(io->ev_flush); io->flushed_since_drain = false; }
        io->drained.store(io->seq.load()); io->off = 0;
    }
    memcpy(io->pinned + io->off, src, size);
    if (io->pinned_dev && size <= MI_IO_DEFER) {
        // deferred: the next graph_compute (or any other reader) moves it with one scatter launch
        io->pending.push_back({ dst, (uint32_t) io->off, (uint32_t) size });
        io->off += need;
        io->seq++;
        return true;
    }
    if (hipMemcpyAsync(dst, io->pinned + io->off, size, hipMemcpyHostToDevice, io->stream) != hipSuccess) return false;
    io->off += need;
    (void) hipEventRecord(io->ev, io->stream);
    io->seq++; io->copy_seq++;
    return true;
}
static bool mi_io_download(int device, void * dst, const void * src, size_t size) {
    mi_io_ctx * io = mi_io(device);
    if (!io || size > io->cap / 2) return false;
    std::lock_guard<std::mutex> lk(io->mtx);
    // the ring is used from its upper half for downloads after draining the stream (uploads in flight keep the lower part)
    mi_io_flush_locked(*io, io->stream);
    (void) hipStreamSynchronize(io->stream);
    if (io->flushed_since_drain) { (void) hipEventSynchronize(io->ev_flush); io->flushed_since_drain = false; }
    io->drained.store(io->seq.load()); io->off = 0;
    char * stage = io->pinned + io->cap / 2;
    if (hipMemcpyAsync(stage, src, size, hipMemcpyDeviceToHost, io->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(io->stream) != hipSuccess) return false;
    memcpy(dst, stage, size);
    return true;
}

// order `cs` behind every input upload accepted so far: deferred uploads (a step's graph inputs) leave with one scatter launch at the
// head of this stream; uploads that another stream flushed, or that went through async copies, are ordered in front of it by their events
struct mi_io_marks;
static void mi_io_order_stream(int device, mi_io_marks & mk, hipStream_t cs);
// host-visible mirrors of logits rows (defined with the backend): stale after any other write to the range, readable once valid
static void mi_mirror_invalidate(int device, const void * p, size_t n);
static bool mi_mirror_read(int device, const void * src, void * dst, size_t size);

// ---------------------------------------------------------------------------------------------------
// buffer
// ---------------------------------------------------------------------------------------------------
static void mi_buffer_free(ggml_backend_buffer_t buffer) {
    mi_buffer_ctx * ctx = (mi_buffer_ctx *) buffer->context;
    {
        std::lock_guard<std::mutex> lk(g_weights_mtx);
        for (size_t i = 0; i < g_buffers.size(); i++) if (g_buffers[i].base == ctx->base) { g_buffers.erase(g_buffers.begin() + i); break; }
    }
    mi_shadows_drop(ctx->device, ctx->base);
    mi_mirror_invalidate(ctx->device, ctx->base, ctx->size);
    (void) hipSetDevice(ctx->device);
    mi_io_drain(ctx->device);
    (void) hipDeviceSynchronize();
    (void) hipFree(ctx->base);
    delete ctx;
}

static void * mi_buffer_get_base(ggml_backend_buffer_t buffer) { return ((mi_buffer_ctx *) buffer->context)->base; }

// quantized tensors are stored planar (include/mi355x_kernels.h); whole-tensor transfers re-layout on the host,
// partial ones go through read-modify-write of the whole tensor (never happens in whisper.cpp: W:1934-1938)
// The planar layout is defined per WHOLE tensor (the planes of NB blocks follow each other): a row / sub-view of a quantized
// tensor has no contiguous image in it, and a non-contiguous one cannot be re-laid out block by block.  whisper.cpp only ever
// transfers whole weight tensors (W:1934-1938); anything else is a programmer error and must not silently corrupt weights.
static bool whole_quant_tensor(const ggml_tensor * t) {
    return ggml_is_contiguous(t) && (!t->view_src || (t->view_offs == 0 && ggml_nbytes(t) == ggml_nbytes(t->view_src)));
}
#define MI_REQUIRE_WHOLE_QUANT(t, what) do { if (is_quant_type((t)->type) && !whole_quant_tensor(t)) \
    GGML_ABORT("ggml-mi355x: %s of a partial / non-contiguous view of quantized tensor '%s' (%s): quantized tensors are stored planar and move as whole tensors only", what, (t)->name, ggml_type_name((t)->type)); } while (0)

static void mi_buffer_set_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    io_timer tm(0);
    mi_buffer_ctx * ctx = (mi_buffer_ctx *) buffer->context;
    (void) hipSetDevice(ctx->device);
    MI_REQUIRE_WHOLE_QUANT(tensor, "set_tensor");
    mi_mirror_invalidate(ctx->device, (const char *) tensor->data + offset, size);
    // (the buffer is not yet marked WEIGHTS while the loader fills it: ggml_backend_buffer_set_usage comes after the loop, W:1956)
    if (t_defer_weights != 0 && buffer->usage != GGML_BACKEND_BUFFER_USAGE_COMPUTE && ggml_nbytes(tensor) >= (1u << 16)) {
        g_deferred_bytes += size;
        MI_LOG("set_tensor of '%s' (%zu bytes) deferred: arrives by broadcast", tensor->name, size);
        return;
    }
    if (is_quant_type(tensor->type)) {
        mi_shadows_drop(ctx->device, ctx->base);
        mi_io_drain(ctx->device);
        const size_t nbytes = ggml_nbytes(tensor);
        std::vector<uint8_t> planar(nbytes);
        if (offset == 0 && size == nbytes) {
            mi355x_repack_to_planar((int) tensor->type, data, planar.data(), ggml_nelements(tensor));
        } else {
            std::vector<uint8_t> blocks(nbytes);
            (void) hipMemcpy(planar.data(), tensor->data, nbytes, hipMemcpyDeviceToHost);
            mi355x_repack_from_planar((int) tensor->type, planar.data(), blocks.data(), ggml_nelements(tensor));
            memcpy(blocks.data() + offset, data, size);
            mi355x_repack_to_planar((int) tensor->type, blocks.data(), planar.data(), ggml_nelements(tensor));
        }
        hipError_t e = hipMemcpy(tensor->data, planar.data(), nbytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) GGML_LOG_ERROR("ggml-mi355x: set_tensor failed: %s\n", hipGetErrorString(e));
        return;
    }
    if (size <= MI_IO_SMALL && mi_io_upload(ctx->device, (char *) tensor->data + offset, data, size)) return;
    mi_io_drain(ctx->device);
    hipError_t e = hipMemcpy((char *) tensor->data + offset, data, size, hipMemcpyHostToDevice);
    if (e != hipSuccess) GGML_LOG_ERROR("ggml-mi355x: set_tensor failed: %s\n", hipGetErrorString(e));
}

static void mi_buffer_get_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    io_timer tm(1);
    mi_buffer_ctx * ctx = (mi_buffer_ctx *) buffer->context;
    if (!is_quant_type(tensor->type) && mi_mirror_read(ctx->device, (const char *) tensor->data + offset, data, size)) {          // logits: already in host memory
        return;
    }
    (void) hipSetDevice(ctx->device);
    mi_io_drain(ctx->device);
    MI_REQUIRE_WHOLE_QUANT(tensor, "get_tensor");
    if (is_quant_type(tensor->type)) {
        const size_t nbytes = ggml_nbytes(tensor);
        std::vector<uint8_t> planar(nbytes), blocks(nbytes);
        (void) hipMemcpy(planar.data(), tensor->data, nbytes, hipMemcpyDeviceToHost);
        mi355x_repack_from_planar((int) tensor->type, planar.data(), blocks.data(), ggml_nelements(tensor));
        memcpy(data, blocks.data() + offset, size);
        return;
    }
    if (size >= 4096 && mi_io_download(ctx->device, data, (const char *) tensor->data + offset, size)) return;
    hipError_t e = hipMemcpy(data, (const char *) tensor->data + offset, size, hipMemcpyDeviceToHost);
    if (e != hipSuccess) GGML_LOG_ERROR("ggml-mi355x: get_tensor failed: %s\n", hipGetErrorString(e));
}

static void mi_buffer_memset_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    mi_buffer_ctx * ctx = (mi_buffer_ctx *) buffer->context;
    MI_REQUIRE_WHOLE_QUANT(tensor, "memset_tensor");
    if (is_quant_type(tensor->type) && !(offset == 0 && size == ggml_nbytes(tensor)))
        GGML_ABORT("ggml-mi355x: partial memset of quantized tensor '%s': planar layout, whole tensors only", tensor->name);
    if (is_quant_type(tensor->type)) mi_shadows_drop(ctx->device, ctx->base);
    mi_mirror_invalidate(ctx->device, (const char *) tensor->data + offset, size);
    (void) hipSetDevice(ctx->device);
    mi_io_drain(ctx->device);
    (void) hipMemset((char *) tensor->data + offset, value, size);
    (void) hipDeviceSynchronize();
}

static void mi_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    mi_buffer_ctx * ctx = (mi_buffer_ctx *) buffer->context;
    mi_shadows_drop(ctx->device, ctx->base);
    mi_mirror_invalidate(ctx->device, ctx->base, ctx->size);
    (void) hipSetDevice(ctx->device);
    mi_io_drain(ctx->device);
    (void) hipMemset(ctx->base, value, ctx->size);
    (void) hipDeviceSynchronize();
}

static bool mi_buffer_is_ours(ggml_backend_buffer_t buffer) { return buffer && buffer->iface.get_base == mi_buffer_get_base; }

static bool mi_buffer_cpy_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * src, ggml_tensor * dst) {
    ggml_backend_buffer_t sbuf = src->view_src ? src->view_src->buffer : src->buffer;
    if (!mi_buffer_is_ours(sbuf)) return false;
    io_timer tm(2);
    mi_buffer_ctx * ctx = (mi_buffer_ctx *) buffer->context;
    MI_REQUIRE_WHOLE_QUANT(src, "cpy_tensor (source)");
    MI_REQUIRE_WHOLE_QUANT(dst, "cpy_tensor (destination)");
    if (is_quant_type(dst->type)) mi_shadows_drop(ctx->device, ctx->base);
    mi_mirror_invalidate(ctx->device, dst->data, ggml_nbytes(dst));
    // uploads still in flight for the SOURCE (it may live on another of our devices) must land before the raw copy reads it
    const int sdev = ((mi_buffer_ctx *) sbuf->context)->device;
    if (sdev != ctx->device) { (void) hipSetDevice(sdev); mi_io_drain(sdev); }
    (void) hipSetDevice(ctx->device);
    mi_io_drain(ctx->device);
    // same layout on both sides (ggml_are_same_layout is asserted by the caller) => raw bytes, planar included
    hipError_t e = hipMemcpy(dst->data, src->data, ggml_nbytes(src), hipMemcpyDeviceToDevice);
    (void) hipDeviceSynchronize();
    return e == hipSuccess;
}

static const ggml_backend_buffer_i mi_buffer_iface = {
    /* .free_buffer   = */ mi_buffer_free,
    /* .get_base      = */ mi_buffer_get_base,
    /* .init_tensor   = */ nullptr,
    /* .memset_tensor = */ mi_buffer_memset_tensor,
    /* .set_tensor    = */ mi_buffer_set_tensor,
    /* .get_tensor    = */ mi_buffer_get_tensor,
    /* .set_tensor_2d = */ nullptr,
    /* .get_tensor_2d = */ nullptr,
    /* .cpy_tensor    = */ mi_buffer_cpy_tensor,
    /* .clear         = */ mi_buffer_clear,
    /* .reset         = */ nullptr,
};

// ---------------------------------------------------------------------------------------------------
// buffer type
// ---------------------------------------------------------------------------------------------------
static const char * mi_buft_get_name(ggml_backend_buffer_type_t buft) { return ((mi_device_ctx *) buft->context)->name.c_str(); }

static ggml_backend_buffer_t mi_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    mi_device_ctx * dev = (mi_device_ctx *) buft->context;
    if (hipSetDevice(dev->index) != hipSuccess) return nullptr;
    void * base = nullptr;
    const size_t asize = size + 1024;          // slack: kernels may read whole 16-byte vectors at the tail
    hipError_t e = hipMalloc(&base, asize);
    if (e != hipSuccess) {
        GGML_LOG_ERROR("ggml-mi355x: hipMalloc of %.2f MiB on device %d failed: %s\n", asize / 1048576.0, dev->index, hipGetErrorString(e));
        return nullptr;
    }
    mi_buffer_ctx * ctx = new mi_buffer_ctx{ dev->index, base, asize };
    ggml_backend_buffer_t buf = ggml_backend_buffer_init(buft, mi_buffer_iface, ctx, size);
    {
        std::lock_guard<std::mutex> lk(g_weights_mtx);
        g_buffers.push_back({ dev->index, base, size, buf });
    }
    return buf;
}
static size_t mi_buft_get_alignment(ggml_backend_buffer_type_t) { return MI_ALIGNMENT; }
static size_t mi_buft_get_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * tensor) { return ggml_nbytes(tensor); }
static bool   mi_buft_is_host(ggml_backend_buffer_type_t) { return false; }

static const ggml_backend_buffer_type_i mi_buft_iface = {
    /* .get_name       = */ mi_buft_get_name,
    /* .alloc_buffer   = */ mi_buft_alloc_buffer,
    /* .get_alignment  = */ mi_buft_get_alignment,
    /* .get_max_size   = */ nullptr,
    /* .get_alloc_size = */ mi_buft_get_alloc_size,
    /* .is_host        = */ mi_buft_is_host,
};

// ---------------------------------------------------------------------------------------------------
// backend (stream)
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// decoder steps with 3..8 columns, and cross-state batches: the pre-quantized-activation pipeline (csrc/kernels/decode_q.hip).
// A stage's activation vector is rounded to the weights' vec_dot_type ONCE (mi355x_act_prepare, or the epilogue of the producing
// mat-vec) and every mat-vec workgroup only copies the planes, so a step costs the same for 1 and for 8 columns.  The columns are
// either the T tokens of ONE graph (beam search: src/whisper.cpp:6486-6543 decodes one token per beam per step) or ONE token of
// each of S graphs — S whisper_states on one device whose single-token steps arrived together (mi_batch_group below): the reference
// batches sequences inside one graph (whisper_batch, src/whisper.cpp:472-523, mask :2928-2945); independent audio streams have no
// common graph, so the batch is formed here, behind the backend boundary.  Per column the arithmetic is that of the fused T <= 2
// kernels, bit for bit (tests/test_gpu_batch.py).
// ---------------------------------------------------------------------------------------------------
struct mi_backend_ctx;
struct mi_colset {
    int S = 1;                                       // 1: the T token columns of one graph (strided); > 1: S graphs, one single-token column each
    int T = 1;                                       // columns in total
    const ggml_cgraph * g[MI355X_MAX_COLS] = {};
    mi_backend_ctx * owner[MI355X_MAX_COLS] = {};    // whose state column c belongs to (S == 1: owner[0] for all) — for the logits mirror
};
struct mi_qstate { const void * src = nullptr; int64_t K = 0; int T = 0; int which = 0; };      // planes a producer's epilogue left for its consumer

struct mi_io_marks { uint64_t seen = 0, copy_seen = 0, flush_seen = 0; };     // uploads (mi_io_ctx::seq / copy_seq / flush_count) a stream already waits behind

struct mi_backend_ctx {
    int          device;
    mi355x_ctx * k;
    std::string  name;
    bool         fuse, prof;
    // GGML_MI355X_EXACT=1: walk the reference CPU path's arithmetic where it differs observably from ours — flash attention in
    // the CPU dispatcher's three forms (F16 accumulation, split over n_threads, F32 tiles; fattn_exact.hip) and integer block dots
    // for every column count (no f16-rounded d*q products) — so that free-running decodes can be compared token for token
    bool         exact = false;
    int          n_threads = 4;
    // f16 activation scratch for the MFMA path, shared by consecutive mul_mats with the same src1
    void *       act = nullptr; size_t act_size = 0;
    void *       act_alt = nullptr; size_t act_alt_size = 0;     // second scratch: a GEMM reading `act` writes the next GEMM's prepared activations here
    const void * act_src = nullptr; int64_t act_K = 0, act_T = 0, act_nb1 = 0; int act_mode = -1;
    const void * elided_src = nullptr;     // F32 result a producer did NOT store because its only reader takes the prepared activations (this graph) ...
    const ggml_tensor * elided_for = nullptr;   // ... that reader: the one node for which reading x->data is an error (the address itself is reused by later tensors)
    mi_io_marks io;                                             // uploads this stream already waits behind
    mi_qstate   qs;                                             // planes a producer's epilogue left for the next mat-vec (T >= 3 pipeline)
    // cross-state batches (mi_batch_group)
    bool        in_group = false;                               // counted among the device's decoding states (guarded by the group's mutex)
    bool        in_flight = false;                              // a column of a chain that is being launched right now (group's mutex)
    bool        own_dirty = false;                              // work was launched on the own stream since the group's stream last waited for it
    hipEvent_t  own_ev = nullptr;
    hipEvent_t  batch_wait_sync = nullptr, batch_wait_stream = nullptr;   // completion of the last batch this state was a column of: synchronize() / the own stream still have to wait for it
    int         no_batch_nodes = 0;                             // graph size (n_nodes) that was found not to fit the batch walker
    uint64_t    sig_nodes = 0; const void * sig_w = nullptr;    // graph shape of THIS state that was last checked congruent with its group's (mi_compute_batch)
    // host-visible mirror of the logits: the vocabulary projection stores its result a second time into pinned, device-mapped host memory,
    // so whisper's read-back of the row(s) (ggml_backend_tensor_get, src/whisper.cpp:2957-2963) is a memcpy instead of a device-to-host copy
    char *      mirror_host = nullptr; char * mirror_dev = nullptr;
    const void * mirror_src = nullptr; size_t mirror_bytes = 0;  // device range [mirror_src, + mirror_bytes) is what the mirror holds
    std::atomic<int> mirror_state{0};                           // 0 nothing, 1 launched (not yet synchronized), 2 valid
    // where the last decoder step of this state left its logits (device): ggml_backend_mi355x_argmax_last reduces a row there
    const float * logits_dev = nullptr; int logits_n = 0, logits_rows = 0;
    uint64_t n_graph_compute = 0;
    double   t_eager_ms = 0;                                    // host time inside graph_compute
    uint64_t trace_gc_enter = 0;                                // GGML_MI355X_TRACE: entry time of the graph_compute still waiting for its synchronize
    // GPU-side span of every graph_compute (first launch .. last kernel done), from a ring of hipEvent pairs on the stream
    std::vector<std::pair<hipEvent_t, hipEvent_t>> span_ev;
    int      span_next = 0, span_pending = 0;
    double   t_gpu_span_ms = 0;
};
static double g_total_gpu_span_ms = 0;

static void mi_span_drain(mi_backend_ctx * b) {          // all pending pairs must have completed (caller synchronized the stream)
    for (int i = 0; i < b->span_pending; i++) {
        const int idx = (b->span_next - 1 - i + 2 * (int) b->span_ev.size()) % (int) b->span_ev.size();
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->span_ev[idx].first, b->span_ev[idx].second) == hipSuccess) b->t_gpu_span_ms += ms;
    }
    b->span_pending = 0;
}

static void mi_io_order_stream(int device, mi_io_marks & mk, hipStream_t cs) {
    mi_io_ctx & io = g_io[device];
    const uint64_t seq = io.ok ? io.seq.load() : 0;
    if (seq == mk.seen) return;
    std::lock_guard<std::mutex> lk(io.mtx);
    if (io.flush_count != mk.flush_seen && io.flush_stream && io.flush_stream != cs) (void) hipStreamWaitEvent(cs, io.ev_flush, 0);
    mi_io_flush_locked(io, cs);
    mk.flush_seen = io.flush_count;
    if (io.copy_seq.load() != mk.copy_seen) { (void) hipStreamWaitEvent(cs, io.ev, 0); mk.copy_seen = io.copy_seq.load(); }
    mk.seen = io.seq.load();
    io.wake_seq = mk.seen;
}

static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static std::vector<mi_backend_ctx *> g_backends;           // live backends (guarded by g_weights_mtx; the mirror look-ups below only share g_backends_rw)
static std::shared_mutex g_backends_rw;                     // writers (backend init / free) hold it exclusively IN ADDITION to g_weights_mtx: every stream's per-step
                                                            // uploads and logits reads scan the list, and must not serialise on one process-wide mutex
static thread_local mi_backend_ctx * t_last_backend = nullptr;    // the backend whose graph_compute this host thread called last (one thread per whisper_state)

#define MI_MIRROR_CAP ((size_t) 2 << 20)
static const bool g_mirror_on = env_flag("GGML_MI355X_LOGITS_MIRROR", true);
static char * mi_mirror_dev(mi_backend_ctx * b) {             // device address of the backend's mirror (allocated on first use), or nullptr
    if (!g_mirror_on) return nullptr;
    if (!b->mirror_host) {
        void * h = nullptr, * d = nullptr;
        // (explicitly coherent: the host reads rows the device wrote, ordered only by an event wait — must hold with HIP_HOST_COHERENT=0 too)
        if (hipHostMalloc(&h, MI_MIRROR_CAP, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void) hipGetLastError(); (void) hipHostFree(h); return nullptr; }
        b->mirror_host = (char *) h; b->mirror_dev = (char *) d;
    }
    return b->mirror_dev;
}
// a write to [p, p + n) of `device` memory that did not come from the mirroring kernel: mirrors of that range are stale
static void mi_mirror_invalidate(int device, const void * p, size_t n) {
    std::shared_lock<std::shared_mutex> lk(g_backends_rw);
    for (auto * b : g_backends)
        if (b->device == device && b->mirror_state.load() != 0 && (const char *) b->mirror_src < (const char *) p + n && (const char *) p < (const char *) b->mirror_src + b->mirror_bytes) b->mirror_state.store(0);
}
// read [src, src + size) from a valid mirror instead of the device; false: no mirror holds it
static bool mi_mirror_read(int device, const void * src, void * dst, size_t size) {
    std::shared_lock<std::shared_mutex> lk(g_backends_rw);          // (shared: several streams copy their rows at the same time)
    for (auto * b : g_backends) {
        if (b->device != device || b->mirror_state.load() != 2) continue;
        const char * s0 = (const char *) b->mirror_src;
        if ((const char *) src >= s0 && (const char *) src + size <= s0 + b->mirror_bytes) { memcpy(dst, b->mirror_host + ((const char *) src - s0), size); return true; }
    }
    return false;
}
static uint64_t g_total_stats[4] = { 0, 0, 0, 0 };         // counters of already freed backends
static double   g_total_host_ms[4] = { 0, 0, 0, 0 };

static mi355x_tensor to_mt(const ggml_tensor * t) {
    mi355x_tensor m;
    m.data = t->data; m.type = (int32_t) t->type; m.reserved = 0;
    for (int i = 0; i < 4; i++) { m.ne[i] = t->ne[i]; m.nb[i] = (int64_t) t->nb[i]; }
    return m;
}

static bool op_is_empty(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE;
}

static int use_count(const ggml_cgraph * g, const ggml_tensor * t) {
    if (!g->use_counts || !g->visited_hash_set.keys) return 1 << 20;
    const size_t pos = ggml_hash_find(&g->visited_hash_set, t);
    if (pos == GGML_HASHSET_FULL || !ggml_bitset_get(g->visited_hash_set.used, pos)) return 1 << 20;
    return g->use_counts[pos];
}
// may `t` be elided (computed only inside a fused kernel) given that exactly `n` fused consumers read it?
static bool can_elide(const ggml_cgraph * g, const ggml_tensor * t, int n) {
    return use_count(g, t) == n && !t->view_src && !(t->flags & GGML_TENSOR_FLAG_OUTPUT);
}

static bool overlap(const void * a, size_t na, const void * b, size_t nb) {
    const char * pa = (const char *) a, * pb = (const char *) b;
    return pa < pb + nb && pb < pa + na;
}
static bool t_overlap(const ggml_tensor * a, const ggml_tensor * b) { return overlap(a->data, ggml_nbytes(a), b->data, ggml_nbytes(b)); }

static bool is_vec_f32(const ggml_tensor * t, int64_t n) {     // contiguous f32 vector of n elements (any trailing 1-dims)
    return t->type == GGML_TYPE_F32 && ggml_nelements(t) == n && t->ne[0] == n && t->nb[0] == 4;
}

// next node index after `i` that is not an empty op (or n_nodes)
static int next_real(const ggml_cgraph * g, int i) {
    int j = i + 1;
    while (j < g->n_nodes && (op_is_empty(g->nodes[j]) || !(g->nodes[j]->flags & GGML_TENSOR_FLAG_COMPUTE))) j++;
    return j;
}

// ---- mul_mat chain:  mul_mat [-> add bias] [-> scale] [-> gelu] [-> add residual] [-> cpy to f16] ----
struct mm_chain {
    const ggml_tensor * mm = nullptr;
    const ggml_tensor * last = nullptr;     // tensor whose memory receives the result
    mi355x_epilogue ep{};
    int end = 0;                            // index of the last fused node
    int res_node = -1, res_slot = 0;        // the residual operand is src[res_slot] of node res_node (cross-state batches look it up per graph)
};

static bool parse_mm_chain(const ggml_cgraph * g, int i, bool fuse, mm_chain & c) {
    const ggml_tensor * mm = g->nodes[i];
    if (mm->op != GGML_OP_MUL_MAT) return false;
    c = mm_chain(); c.mm = mm; c.last = mm; c.end = i;
    if (!fuse) return true;
    const ggml_tensor * w = mm->src[0], * x = mm->src[1];
    if (ggml_n_dims(w) > 2 || x->ne[2] != 1 || x->ne[3] != 1 || mm->type != GGML_TYPE_F32) return true;
    const int64_t N = mm->ne[0];
    const ggml_tensor * cur = mm;
    int stage = 0;   // 0: bias allowed, 1: scale, 2: gelu, 3: residual, 4: cpy
    int j = next_real(g, i);
    // ggml_conv_1d (ggml.c:4537-4554) puts a RESHAPE between its mul_mat and the bias add (src/whisper.cpp:2013-2020): the add reads the
    // product through a same-shape contiguous view, and its bias has one value per COLUMN of the product ([1, OC] against [OL, OC])
    if (j < g->n_nodes && g->nodes[j]->op == GGML_OP_ADD && can_elide(g, mm, 1) && mm->ne[2] == 1 && mm->ne[3] == 1) {
        const ggml_tensor * n = g->nodes[j];
        for (int sl = 0; sl < 2; sl++) {
            const ggml_tensor * v = n->src[sl], * o = n->src[1 - sl];
            if (v->op == GGML_OP_RESHAPE && v->src[0] == mm && v->data == mm->data && ggml_are_same_shape(v, mm) && ggml_is_contiguous(v) &&
                use_count(g, v) == 1 && !(v->flags & GGML_TENSOR_FLAG_OUTPUT) && ggml_are_same_shape(n, mm) && n->type == GGML_TYPE_F32 &&
                o->type == GGML_TYPE_F32 && o->ne[0] == 1 && o->ne[1] == mm->ne[1] && ggml_nelements(o) == mm->ne[1] && o->nb[1] == 4 && mm->ne[1] > 8) {
                c.ep.bias = (const float *) o->data; c.ep.bias_per_col = 1;
                stage = 1; cur = n; c.last = n; c.end = j;
                j = next_real(g, j);
                break;
            }
        }
    }
    while (j < g->n_nodes && stage < 5) {
        const ggml_tensor * n = g->nodes[j];
        if (!can_elide(g, cur, 1)) break;
        bool took = false;
        if (n->op == GGML_OP_ADD && (n->src[0] == cur || n->src[1] == cur) && ggml_are_same_shape(n, cur) && n->type == GGML_TYPE_F32) {
            const ggml_tensor * o = n->src[0] == cur ? n->src[1] : n->src[0];
            if (stage <= 0 && is_vec_f32(o, N) && o != cur) { c.ep.bias = (const float *) o->data; stage = 1; took = true; }
            else if (stage <= 3 && o->type == GGML_TYPE_F32 && ggml_are_same_shape(o, cur) && o->nb[0] == 4 && o != cur) {
                c.ep.residual = (const float *) o->data; c.ep.residual_nb1 = (int64_t) o->nb[1]; stage = 4; took = true;
                c.res_node = j; c.res_slot = n->src[0] == cur ? 1 : 0;
            }
        } else if (n->op == GGML_OP_SCALE && n->src[0] == cur && stage <= 1 && ggml_get_op_params_f32(n, 1) == 0.0f) {
            c.ep.scale = ggml_get_op_params_f32(n, 0); c.ep.has_scale = 1; stage = 2; took = true;
        } else if (n->op == GGML_OP_UNARY && ggml_get_unary_op(n) == GGML_UNARY_OP_GELU && n->src[0] == cur && stage <= 2) {
            c.ep.gelu = 1; stage = 3; took = true;
        } else if (n->op == GGML_OP_CPY && n->src[0] == cur && n->type == GGML_TYPE_F16 && ggml_is_contiguous(n) &&
                   ggml_nelements(n) == ggml_nelements(cur) && ggml_is_contiguous(cur)) {
            stage = 5; took = true;
        }
        if (!took) break;
        cur = n; c.last = n; c.end = j;
        j = next_real(g, j);
    }
    // memory hazards: the result must not land on anything the kernel still reads
    const ggml_tensor * res_t = nullptr;
    if (c.last != mm) {
        bool bad = t_overlap(c.last, x) || t_overlap(c.last, w);
        if (c.ep.residual) {
            // identical aliasing (in-place add) is fine: every element is read before it is written by the same lane
            const char * r = (const char *) c.ep.residual;
            const size_t rn = (size_t) c.ep.residual_nb1 * (size_t) mm->ne[1];
            if (overlap(c.last->data, ggml_nbytes(c.last), r, rn) && !(r == (const char *) c.last->data && c.ep.residual_nb1 == (int64_t) c.last->nb[1])) bad = true;
        }
        if (c.ep.bias && overlap(c.last->data, ggml_nbytes(c.last), c.ep.bias, (c.ep.bias_per_col ? mm->ne[1] : N)*4)) bad = true;
        (void) res_t;
        if (bad) { c = mm_chain(); c.mm = mm; c.last = mm; c.end = i; }
    }
    return true;
}

static int mode_for(ggml_type t) { return t == GGML_TYPE_Q4_K ? 2 : (is_quant_type(t) ? 1 : 0); }
// The int8 tile GEMM over the quantized operands (csrc/kernels/mmq.hip) takes every product of a quantized weight with more than 8
// columns whose K it can tile: its activations are the reference's Q8_0 / Q8_K blocks as integers ("rows", prep modes 3 / 4) and its
// A operand the planar quantized weight itself — no f16 copy of a weight is made or read.  GGML_MI355X_MMQ=0 brings back the f16 MFMA
// path (f16(d*q) activations, f16 weight copies).
static bool mi_mmq_on() { static const bool on = env_flag("GGML_MI355X_MMQ", true); return on; }
static int rows_mode_for(const ggml_tensor * w, int64_t K) {
    if (!mi_mmq_on() || !is_quant_type(w->type) || K % 128 != 0) return 0;
    return w->type == GGML_TYPE_Q4_K ? (K % 256 == 0 ? 4 : 0) : 3;
}

// f16 copy of a quantized weight for the MFMA path (nullptr: not eligible / over budget -> the GEMM dequantizes in its loop)
static const void * mi_shadow_get(mi_backend_ctx * b, const ggml_tensor * w, const mi355x_tensor & mw) {
    constexpr size_t cap_mb = 16384;               // f16 copies of quantized weights (GGML_MI355X_MMQ=0 only): at most 16 GB of the 288
    if (cap_mb == 0) return nullptr;
    ggml_backend_buffer_t buf = w->view_src ? w->view_src->buffer : w->buffer;
    if (!buf || !mi_buffer_is_ours(buf) || buf->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS) return nullptr;
    std::lock_guard<std::mutex> lk(g_shadow_mtx);
    auto it = g_shadows.find(w->data);
    if (it != g_shadows.end()) {
        const mi_shadow & sh = it->second;
        return (sh.type == (int) w->type && sh.ne0 == w->ne[0] && sh.ne1 == w->ne[1]) ? sh.f16 : nullptr;
    }
    const size_t bytes = (size_t) w->ne[0] * (size_t) w->ne[1] * 2;
    if (g_shadow_bytes + bytes > cap_mb * 1024 * 1024) return nullptr;
    const mi_buffer_ctx * bc = (const mi_buffer_ctx *) buf->context;
    void * p = nullptr;
    if (hipMalloc(&p, bytes + 256) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    // the copy is published only once it is complete, so another backend (another whisper_state on its own stream) never reads it early
    if (mi355x_dequant_f16(b->k, &mw, p) != 0 || mi355x_ctx_synchronize(b->k) != 0) { (void) hipFree(p); return nullptr; }
    g_shadows[w->data] = { p, bytes, bc->base, bc->device, (int) w->type, w->ne[0], w->ne[1] };
    g_shadow_bytes += bytes;
    g_shadow_count.store(g_shadows.size());
    return p;
}

// room for T x K prepared f16 activations in the backend's scratch; 0 ok, > 0 error
static int mi_act_reserve(mi_backend_ctx * b, size_t need, bool alt = false) {
    void * & buf = alt ? b->act_alt : b->act;
    size_t & size = alt ? b->act_alt_size : b->act_size;
    if (need <= size) return 0;
    mi355x_ctx_synchronize(b->k);                 // (also sends held-back launches that still name the old buffer)
    if (buf) (void) hipFree(buf);
    buf = nullptr; size = 0;
    if (!alt) b->act_src = nullptr;
    const size_t sz = need + (need >> 2);
    if (hipMalloc(&buf, sz) != hipSuccess) return (int) hipErrorOutOfMemory;
    size = sz;
    return 0;
}

// does tensor x (F32 [K, T]) feed the MFMA GEMM path as the activation of mul_mat `mm`?  (the conditions of run_mm_chain)
static bool mm_takes_prepared(const mi_backend_ctx * b, const ggml_tensor * mm, const ggml_tensor * x, int & mode_out) {
    if (mm->op != GGML_OP_MUL_MAT || mm->src[1] != x || b->exact) return false;
    const ggml_tensor * w = mm->src[0];
    const int64_t K = w->ne[0], T = x->ne[1];
    const bool two_d = ggml_n_dims(w) <= 2 && x->ne[2] == 1 && x->ne[3] == 1;
    if (!(two_d && T > 8 && K % 8 == 0 && x->type == GGML_TYPE_F32 && x->nb[0] == 4 && (x->nb[1] % 16 == 0) && ((uintptr_t) x->data % 16 == 0))) return false;
    if (!((is_quant_type(w->type) && ggml_is_contiguous(w)) || (w->type == GGML_TYPE_F16 && w->nb[0] == 2 && w->nb[1] % 16 == 0))) return false;
    const int mode = mode_for(w->type);
    if ((mode == 1 && K % 32) || (mode == 2 && K % 256)) return false;
    int rmode = rows_mode_for(w, K);
    if (rmode && ((uintptr_t) w->data % 16)) rmode = 0;                   // (mi355x_gemm_q8act's own precondition: never promise rows it would refuse)
    mode_out = rmode ? rmode : mode;
    return true;
}

static int run_mm_chain(mi_backend_ctx * b, const mm_chain & c, const ggml_cgraph * g = nullptr) {
    const ggml_tensor * mm = c.mm, * w = mm->src[0], * x = mm->src[1];
    // a producer elided this F32 activation because mm_takes_prepared() promised that its prepared rows would be consumed.  If the
    // consuming kernel then refuses them (alignment, a shape only its launch code knows), no path that re-reads x->data may run: fail loudly.
    auto reads_elided = [&]() {
        if (!b->elided_src || mm != b->elided_for || x->data != b->elided_src) return false;
        GGML_LOG_ERROR("ggml-mi355x: %s: the prepared activations of %s were refused and its F32 form was never stored\n", mm->name, x->name);
        return true;
    };
    mi355x_tensor mw = to_mt(w), mx = to_mt(x);
    // destination: the chain's last tensor, seen as [N, T] with the dtype of that tensor
    mi355x_tensor md = to_mt(mm);
    md.data = c.last->data; md.type = (int32_t) c.last->type;
    if (c.last->type == GGML_TYPE_F16) { md.nb[0] = 2; md.nb[1] = mm->ne[0]*2; md.nb[2] = md.nb[1]*mm->ne[1]; md.nb[3] = md.nb[2]; }
    else if (c.last != mm)              { md.nb[0] = 4; md.nb[1] = (int64_t) c.last->nb[1]; md.nb[2] = (int64_t) c.last->nb[2]; md.nb[3] = (int64_t) c.last->nb[3]; }
    const bool has_ep = c.ep.bias || c.ep.has_scale || c.ep.gelu || c.ep.residual;
    const int64_t K = w->ne[0], T = x->ne[1];
    const bool two_d = ggml_n_dims(w) <= 2 && x->ne[2] == 1 && x->ne[3] == 1;

    if (b->exact && two_d && T > 8 && is_quant_type(w->type) && x->type == GGML_TYPE_F32) {
        // reference-exact mode: the integer-dot mat-vec kernels on 8-column slices (same integer sums as the CPU's vec_dot, f32
        // scale-accumulate) instead of the MFMA path whose activations are f16-rounded d*q products
        for (int64_t t0 = 0; t0 < T; t0 += 8) {
            const int64_t nt = std::min<int64_t>(8, T - t0);
            mi355x_tensor sx = mx, sd = md;
            sx.data = (char *) mx.data + t0 * mx.nb[1]; sx.ne[1] = nt;
            sd.data = (char *) md.data + t0 * md.nb[1]; sd.ne[1] = nt;
            mi355x_epilogue ep = c.ep;
            if (ep.residual) ep.residual = (const float *) ((const char *) ep.residual + t0 * ep.residual_nb1);
            if (ep.bias && ep.bias_per_col) ep.bias += t0;
            const int rc = mi355x_mul_mat(b->k, &mw, &sx, &sd, has_ep ? &ep : nullptr);
            if (rc) return rc;
        }
        return 0;
    }

    // MFMA path with shared prepared activation
    if (two_d && T > 8 && K % 8 == 0 && x->nb[0] == ggml_type_size(x->type) && (x->nb[1] % 16 == 0) && ((uintptr_t) x->data % 16 == 0) &&
        (x->type == GGML_TYPE_F32 || x->type == GGML_TYPE_F16) &&
        ((is_quant_type(w->type) && ggml_is_contiguous(w)) || (w->type == GGML_TYPE_F16 && w->nb[0] == 2 && w->nb[1] % 16 == 0))) {
        const int mode = mode_for(w->type);
        const int rmode = rows_mode_for(w, K);
        if (rmode && ggml_is_contiguous(w)) {
            // the int8 tile GEMM on the quantized weight and the activation rows
            const int rr = mi_act_reserve(b, (size_t) T * K * 2);          // (rows need 1.25 bytes per element: the f16 size covers them)
            if (rr != 0) return rr;
            if (!(b->act_src == x->data && b->act_K == K && b->act_T == T && b->act_mode == rmode && b->act_nb1 == (int64_t) x->nb[1])) {
                if (reads_elided()) return (int) hipErrorInvalidValue;
                const int rc = mi355x_prep_act(b->k, x->data, (int64_t) x->nb[1], x->type == GGML_TYPE_F16, b->act, (int) K, T, rmode);
                if (rc && rc != MI355X_E_UNSUPPORTED) return rc;
                if (rc == 0) { b->act_src = x->data; b->act_K = K; b->act_T = T; b->act_mode = rmode; b->act_nb1 = (int64_t) x->nb[1]; }
                else b->act_src = nullptr;
            }
            if (b->act_src == x->data && b->act_mode == rmode) {
                // fc1 + GELU -> fc2: the epilogue writes the next product's rows (second scratch), and the F32 result only if somebody reads it
                const int64_t M = mm->ne[0];
                int rc = MI355X_E_UNSUPPORTED;
                if (g && b->fuse && c.last->type == GGML_TYPE_F32 && M % 128 == 0 && M <= 8192 &&
                    c.last->nb[0] == 4 && (int64_t) c.last->nb[1] == M*4 && c.last->ne[1] == T && c.last->ne[2] == 1 && c.last->ne[3] == 1) {
                    const int j = next_real(g, c.end);
                    int mode2 = -1;
                    if (j < g->n_nodes && mm_takes_prepared(b, g->nodes[j], c.last, mode2) && mode2 == 3 && mi_act_reserve(b, (size_t) T * M * 2, true) == 0) {
                        const bool only = can_elide(g, c.last, 1);
                        rc = mi355x_gemm_q8act_prep(b->k, &mw, b->act, T, only ? nullptr : md.data, md.nb[1], has_ep ? &c.ep : nullptr, b->act_alt);
                        if (rc == 0) {
                            std::swap(b->act, b->act_alt); std::swap(b->act_size, b->act_alt_size);
                            b->act_src = c.last->data; b->act_K = M; b->act_T = T; b->act_mode = 3; b->act_nb1 = M*4;
                            if (only) { b->elided_src = c.last->data; b->elided_for = g->nodes[j]; }
                            return 0;
                        }
                        if (rc != MI355X_E_UNSUPPORTED) return rc;
                    }
                }
                rc = mi355x_gemm_q8act(b->k, &mw, b->act, T, md.data, md.nb[1], md.type, has_ep ? &c.ep : nullptr);
                if (rc != MI355X_E_UNSUPPORTED) return rc;
            }
        }
        if (!((mode == 1 && K % 32) || (mode == 2 && K % 256))) {
            const void * act; int64_t ld;
            if (x->type == GGML_TYPE_F16 && mode == 0) { act = x->data; ld = (int64_t) x->nb[1] / 2; }
            else {
                const int rr = mi_act_reserve(b, (size_t) T * K * 2);
                if (rr != 0) return rr;
                if (!(b->act_src == x->data && b->act_K == K && b->act_T == T && b->act_mode == mode && b->act_nb1 == (int64_t) x->nb[1])) {
                    if (reads_elided()) return (int) hipErrorInvalidValue;        // (the prepared form on hand is not this path's: x->data would be read)
                    const int rc = mi355x_prep_act(b->k, x->data, (int64_t) x->nb[1], x->type == GGML_TYPE_F16, b->act, (int) K, T, mode);
                    if (rc) return rc;
                    b->act_src = x->data; b->act_K = K; b->act_T = T; b->act_mode = mode; b->act_nb1 = (int64_t) x->nb[1];
                }
                act = b->act; ld = K;
            }
            // Is the result itself the activation matrix of the next node's MFMA GEMM (fc1 + GELU -> fc2)?  Then the epilogue writes
            // that GEMM's prepared f16 activations (second scratch; this product still reads the first), and when nothing else reads
            // the F32 result it is not stored at all: one launch, a 30 MB write and a 30 MB read less per encoder layer of large-v3.
            void * prep_out = nullptr; bool prep_only = false; const ggml_tensor * prep_for = nullptr;
            const int64_t M = mm->ne[0];
            if (g && b->fuse && act == b->act && c.last->type == GGML_TYPE_F32 && M % 32 == 0 && M <= 8192 &&
                c.last->nb[0] == 4 && (int64_t) c.last->nb[1] == M*4 && c.last->ne[1] == T && c.last->ne[2] == 1 && c.last->ne[3] == 1) {
                const int j = next_real(g, c.end);
                int mode2 = -1;
                if (j < g->n_nodes && mm_takes_prepared(b, g->nodes[j], c.last, mode2) && mode2 == 1 && mi_act_reserve(b, (size_t) T * M * 2, true) == 0) {
                    prep_out = b->act_alt; prep_for = g->nodes[j];
                    prep_only = can_elide(g, c.last, 1);
                }
            }
            auto gemm = [&](const mi355x_tensor & wt) {
                if (prep_out) {
                    const int rc = mi355x_gemm_f16act_prep(b->k, &wt, act, ld, T, prep_only ? nullptr : md.data, md.nb[1], has_ep ? &c.ep : nullptr, prep_out);
                    if (rc == 0) {       // the next GEMM finds its activations prepared: the scratch buffers trade places
                        std::swap(b->act, b->act_alt); std::swap(b->act_size, b->act_alt_size);
                        b->act_src = c.last->data; b->act_K = M; b->act_T = T; b->act_mode = 1; b->act_nb1 = M*4;
                        if (prep_only) { b->elided_src = c.last->data; b->elided_for = prep_for; }
                        return 0;
                    }
                    if (rc != MI355X_E_UNSUPPORTED) return rc;
                }
                return mi355x_gemm_f16act(b->k, &wt, act, ld, T, md.data, md.nb[1], md.type, has_ep ? &c.ep : nullptr);
            };
            // wide activations: run the GEMM on the weight's f16 copy (same values, no dequantization in the loop)
            constexpr int shadow_min_t = 128;
            if (mode != 0 && T >= shadow_min_t) {
                if (const void * f16 = mi_shadow_get(b, w, mw)) {
                    mi355x_tensor ms = mw;
                    ms.data = (void *) f16; ms.type = MI355X_TYPE_F16;
                    ms.nb[0] = 2; ms.nb[1] = K*2; ms.nb[2] = ms.nb[1]*w->ne[1]; ms.nb[3] = ms.nb[2];
                    const int rc = gemm(ms);
                    if (rc != MI355X_E_UNSUPPORTED) return rc;
                }
            }
            const int rc = gemm(mw);
            if (rc != MI355X_E_UNSUPPORTED) return rc;
        }
    }
    if (reads_elided()) return (int) hipErrorInvalidValue;                // the generic kernel reads x->data
    return mi355x_mul_mat(b->k, &mw, &mx, &md, has_ep ? &c.ep : nullptr);
}

// ---- norm [-> mul w -> add b] ---------------------------------------------------------------------
struct ln_chain { const ggml_tensor * norm = nullptr, * last = nullptr; const float * w = nullptr, * b = nullptr; int end = 0; };

static void parse_ln_chain(const ggml_cgraph * g, int i, bool fuse, ln_chain & c) {
    const ggml_tensor * nrm = g->nodes[i];
    c = ln_chain(); c.norm = nrm; c.last = nrm; c.end = i;
    if (!fuse || nrm->type != GGML_TYPE_F32) return;
    const int64_t n = nrm->ne[0];
    int j = next_real(g, i);
    if (j >= g->n_nodes || !can_elide(g, nrm, 1)) return;
    const ggml_tensor * m = g->nodes[j];
    if (m->op != GGML_OP_MUL || !ggml_are_same_shape(m, nrm)) return;
    const ggml_tensor * wv = m->src[0] == nrm ? m->src[1] : (m->src[1] == nrm ? m->src[0] : nullptr);
    if (!wv || !is_vec_f32(wv, n)) return;
    int j2 = next_real(g, j);
    if (j2 >= g->n_nodes || !can_elide(g, m, 1)) return;
    const ggml_tensor * a = g->nodes[j2];
    if (a->op != GGML_OP_ADD || !ggml_are_same_shape(a, m)) return;
    const ggml_tensor * bv = a->src[0] == m ? a->src[1] : (a->src[1] == m ? a->src[0] : nullptr);
    if (!bv || !is_vec_f32(bv, n)) return;
    const ggml_tensor * x = nrm->src[0];
    // result memory may alias x exactly (row-wise in-place), but must not partially overlap it
    if (t_overlap(a, x) && !(a->data == x->data && a->nb[1] == x->nb[1] && a->nb[2] == x->nb[2] && a->nb[3] == x->nb[3])) return;
    if (t_overlap(a, wv) || t_overlap(a, bv)) return;
    c.last = a; c.w = (const float *) wv->data; c.b = (const float *) bv->data; c.end = j2;
}

static int run_ln_chain(mi_backend_ctx * b, const ln_chain & c, const ggml_cgraph * g = nullptr) {
    float eps; memcpy(&eps, c.norm->op_params, sizeof(float));
    mi355x_tensor mx = to_mt(c.norm->src[0]), md = to_mt(c.last);
    // encoder / prompt: the next node is an MFMA GEMM on this LayerNorm's result -> write its prepared f16 activations in the same
    // pass (one launch and one read of the 7.7 MB result less per LayerNorm; bit-identical to mi355x_prep_act on the result)
    constexpr bool fuse_prep = true;
    int mode = 0;
    const int j = g ? next_real(g, c.end) : 0;
    if (g && b->fuse && fuse_prep && j < g->n_nodes && mm_takes_prepared(b, g->nodes[j], c.last, mode)) {
        const int64_t K = c.last->ne[0], T = c.last->ne[1];
        const int rr = mi_act_reserve(b, (size_t) T * K * 2);
        if (rr > 0) return rr;
        if (rr == 0) {
            const int rc = mi355x_norm_prep(b->k, &mx, &md, eps, c.w, c.b, b->act, mode);
            if (rc == 0) { b->act_src = c.last->data; b->act_K = K; b->act_T = T; b->act_mode = mode; b->act_nb1 = (int64_t) c.last->nb[1]; return 0; }
            if (rc != MI355X_E_UNSUPPORTED) return rc;
        }
    }
    b->act_src = nullptr;
    return mi355x_norm(b->k, &mx, &md, eps, c.w, c.b);
}

static void attn_consume(mi_backend_ctx * b, const ggml_cgraph * g, int i, const mi355x_attn_partials & parts, int & end_out, int & rc_out);
static int  run_node(mi_backend_ctx * b, const ggml_tensor * n);

// TEST fault injection (negative control of the parity tests: a test that cannot fail proves nothing).  GGML_MI355X_TEST_FAULT=
// "xattn:<n>:<factor>" multiplies the output of the n-th cross-attention block of a decoder step (n = -1: all of them) — the
// (W_o . attention + bias) that is added to the residual stream, src/whisper.cpp:2703-2770 — by `factor`, for steps of up to 8 columns.
// Cross-attention = the FLASH_ATTN_EXT nodes without a mask (decoder self-attention carries one; encoder attention has > 8 columns and
// never passes here).  Unset: the factor is exactly 1 and nothing changes.
// "reject:<n>": the n-th merged launch chain of the process (cross-state batching, counted from 1) reports a kernel-side rejection half-way
// through its walk — the mid-chain path of mi_compute_batch (drain, every member repeats the step on its own chain, the shape stops batching).
struct mi_test_fault { int layer = -2; float factor = 1.0f; int reject_chain = 0; };
static const mi_test_fault & mi_fault() {
    static const mi_test_fault f = [] {
        mi_test_fault t;
        const char * e = getenv("GGML_MI355X_TEST_FAULT");
        if (e && !strncmp(e, "xattn:", 6)) { int l = 0; float x = 1.0f; if (sscanf(e + 6, "%d:%f", &l, &x) == 2) { t.layer = l; t.factor = x; } }
        if (e && !strncmp(e, "reject:", 7)) t.reject_chain = atoi(e + 7);
        return t;
    }();
    return f;
}
// epilogue scale of the output projection that consumes FLASH_ATTN_EXT node i of graph g (1 = untouched)
static float mi_fault_scale(const ggml_cgraph * g, int i) {
    const mi_test_fault & f = mi_fault();
    if (f.layer == -2 || g->nodes[i]->src[3]) return 1.0f;
    int ord = 0;
    for (int j = 0; j < i; j++) if (g->nodes[j]->op == GGML_OP_FLASH_ATTN_EXT && !g->nodes[j]->src[3]) ord++;
    return (f.layer < 0 || f.layer == ord) ? f.factor : 1.0f;
}
static void mi_fault_apply(const ggml_cgraph * g, int i, mi355x_epilogue & ep) {
    const float fs = mi_fault_scale(g, i);
    if (fs != 1.0f) { ep.scale = (ep.has_scale ? ep.scale : 1.0f) * fs; ep.has_scale = 1; }
}


// is this chain the vocabulary projection whose rows the caller reads back (src/whisper.cpp:2957-2963)?  Then its rows are mirrored.
// (whisper does not flag the logits as a graph output; they are the LAST node of the decoder graph, src/whisper.cpp:2827-2840)
static bool mirror_wanted(const ggml_cgraph * g, const mm_chain & ch, int64_t T) {
    const ggml_tensor * l = ch.last;
    return g_mirror_on && ch.end == g->n_nodes - 1 && l == ch.mm && l->type == GGML_TYPE_F32 && l->ne[0] > 8192 && (int64_t) l->nb[1] == l->ne[0]*4 &&
           l->ne[2] == 1 && l->ne[3] == 1 && T >= 1 && T <= MI355X_MAX_COLS && (size_t) (l->ne[0]*4*T) <= MI_MIRROR_CAP - 64;
}

// decoder step: LayerNorm fused into the mat-vec products that consume it (Q/K/V, cross-Q, fc1)
static bool try_ln_gemv(mi_backend_ctx * b, const ggml_cgraph * g, const ln_chain & ln, int & end_out, int & rc_out) {
    if (!ln.w || !ln.b) return false;
    const ggml_tensor * x = ln.norm->src[0], * lnout = ln.last;
    const int64_t K = x->ne[0], T = ggml_nrows(x);
    if (T > 8 || K > 2048 || K % 4 || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[0] != 4 || (x->nb[1] % 16) || ((uintptr_t) x->data % 16)) return false;
    if (((uintptr_t) ln.w % 16) || ((uintptr_t) ln.b % 16)) return false;
    const int nuse = use_count(g, lnout);
    if (nuse < 1 || nuse > 3 || lnout->view_src || (lnout->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    mm_chain ch[3];
    int j = next_real(g, ln.end), n = 0;
    while (n < nuse && j < g->n_nodes) {
        const ggml_tensor * t = g->nodes[j];
        if (t->op != GGML_OP_MUL_MAT || t->src[1] != lnout) return false;
        if (!parse_mm_chain(g, j, true, ch[n])) return false;
        j = next_real(g, ch[n].end);
        n++;
    }
    if (n != nuse) return false;
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    d.x = (const float *) x->data; d.x_nb1 = (int64_t) x->nb[1]; d.K = (int) K; d.T = (int) T;
    d.has_norm = 1; memcpy(&d.eps, ln.norm->op_params, sizeof(float)); d.ln_w = ln.w; d.ln_b = ln.b; d.nseg = n;
    for (int s = 0; s < n; s++) {
        const ggml_tensor * w = ch[s].mm->src[0];
        if (w->type != ch[0].mm->src[0]->type || ggml_n_dims(w) > 2 || w->ne[0] != K) return false;
        if (!((is_quant_type(w->type) && ggml_is_contiguous(w)) || (w->type == GGML_TYPE_F16 && ggml_is_contiguous(w)))) return false;
        if (t_overlap(ch[s].last, x)) return false;
        for (int s2 = 0; s2 < s; s2++) if (t_overlap(ch[s].last, ch[s2].last)) return false;
        mi355x_gemv_seg & sg = d.seg[s];
        sg.w = w->data; sg.wtype = (int32_t) w->type; sg.N = (int32_t) w->ne[1]; sg.ep = ch[s].ep;
        sg.dst = ch[s].last->data; sg.dst_type = (int32_t) ch[s].last->type;
        sg.dst_nb1 = ch[s].last->type == GGML_TYPE_F16 ? (int64_t) w->ne[1]*2 : (ch[s].last == ch[s].mm ? (int64_t) ch[s].mm->nb[1] : (int64_t) ch[s].last->nb[1]);
    }
    // (A one-launch "LN + Q projection + cross-attention" kernel existed in round 1.  Re-measured with plain launches it LOSES to the
    //  two launches it replaced, 1.478 -> 1.435 ms/token with it switched off (profiles/r02_decode_env_sweep_final.txt): 64 rows of
    //  W_q per workgroup serialise what 256 workgroups otherwise do in parallel, and a dependent boundary costs only ~1.5 us.  Removed.)
    // (r03 experiment, removed again: LayerNorm + Q/K/V + cache stores + the flash_attn_ext that follows as ONE launch with a workgroup
    //  per head — no hand-off between workgroups, bit-identical records, one dependent launch less per layer.  It LOST: 13.9 us against
    //  4.4 + 4.3 us stand-alone, 351-353 vs 349 ms per chunk (profiles/r03b_head_kbench.txt, r03b_head_check.txt): a head's 192 rows of
    //  integer dots are VALU-bound on ONE CU, ~3 us where 240 workgroups need 0.2.  The kernel is commit 0693a28.)
    mi355x_gemv_cols mcols;
    if (n == 1 && mirror_wanted(g, ch[0], T)) {
        if (char * md = mi_mirror_dev(b)) {
            memset(&mcols, 0, sizeof(mcols));
            for (int t = 0; t < (int) T; t++) mcols.mirror[t] = md + (size_t) t * (size_t) ch[0].last->ne[0] * 4;
            d.cols = &mcols;
        }
    }
    const int rc = mi355x_gemv_fused(b->k, &d);
    if (rc == MI355X_E_UNSUPPORTED) return false;
    if (rc == 0 && d.cols && mi355x_last_launch_mirrored(b->k)) {
        b->mirror_src = ch[0].last->data; b->mirror_bytes = (size_t) ch[0].last->ne[0] * 4 * (size_t) T; b->mirror_state.store(1);
    }
    rc_out = rc; end_out = ch[n - 1].end;
    return true;
}

static void attn_consume(mi_backend_ctx * b, const ggml_cgraph * g, int i, const mi355x_attn_partials & parts, int & end_out, int & rc_out);

// decoder step: flash_attn_ext (T <= 8) -> reshape -> mul_mat chain (the O-projection).  The attention kernel leaves
// per-128-key partial records; their combine runs in the prologue of the projection mat-vec (one kernel less per
// attention, src/whisper.cpp:2623-2660 and :2703-2770)
static bool try_fattn_gemv(mi_backend_ctx * b, const ggml_cgraph * g, int i, int & end_out, int & rc_out) {
    const ggml_tensor * fa = g->nodes[i];
    const ggml_tensor * q = fa->src[0], * k = fa->src[1], * v = fa->src[2], * m = fa->src[3];
    const int64_t T = q->ne[1];
    if (T > 8 || q->ne[3] != 1 || fa->type != GGML_TYPE_F32 || !ggml_is_contiguous(fa)) return false;
    mi355x_tensor mq = to_mt(q), mk = to_mt(k), mv = to_mt(v), mm_;
    if (m) mm_ = to_mt(m);
    float scale; memcpy(&scale, fa->op_params, 4);
    mi355x_attn_partials parts;
    int rc = mi355x_flash_attn_partial(b->k, &mq, &mk, &mv, m ? &mm_ : nullptr, scale, &parts);
    if (rc == MI355X_E_UNSUPPORTED) return false;
    rc_out = rc; end_out = i;
    if (rc) return true;
    attn_consume(b, g, i, parts, end_out, rc_out);
    return true;
}

// the partial records of flash_attn_ext node i exist: either the projection that follows consumes them (combine in its
// prologue), or they are combined into the node's own memory
static void attn_consume(mi_backend_ctx * b, const ggml_cgraph * g, int i, const mi355x_attn_partials & parts, int & end_out, int & rc_out) {
    const ggml_tensor * fa = g->nodes[i];
    const int64_t T = fa->src[0]->ne[1], H = fa->src[0]->ne[2];
    int rc;
    end_out = i;
    bool fused = false;
    const int j = next_real(g, i);
    mm_chain ch;
    if (j < g->n_nodes && g->nodes[j]->op == GGML_OP_MUL_MAT && can_elide(g, fa, 1)) {
        const ggml_tensor * mm = g->nodes[j], * x = mm->src[1], * w = mm->src[0];
        const bool x_is_fa = x->view_src == fa && x->view_offs == 0 && x->type == GGML_TYPE_F32 && ggml_is_contiguous(x) &&
                             x->ne[0] == H*64 && x->ne[1] == T && x->ne[2] == 1 && x->ne[3] == 1 &&
                             use_count(g, x) == 1 && !(x->flags & GGML_TENSOR_FLAG_OUTPUT);
        if (x_is_fa && is_quant_type(w->type) && ggml_is_contiguous(w) && ggml_n_dims(w) <= 2 && w->ne[0] == H*64 &&
            parse_mm_chain(g, j, true, ch)) {
            mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
            d.K = (int) (H*64); d.T = (int) T; d.nseg = 1;
            d.attn_part_o = parts.part_o; d.attn_part_ml = parts.part_ml; d.attn_nparts = parts.nparts;
            mi355x_gemv_seg & sg = d.seg[0];
            sg.w = w->data; sg.wtype = (int32_t) w->type; sg.N = (int32_t) w->ne[1]; sg.ep = ch.ep;
            mi_fault_apply(g, i, sg.ep);
            sg.dst = ch.last->data; sg.dst_type = (int32_t) ch.last->type;
            sg.dst_nb1 = ch.last->type == GGML_TYPE_F16 ? (int64_t) w->ne[1]*2 : (ch.last == ch.mm ? (int64_t) ch.mm->nb[1] : (int64_t) ch.last->nb[1]);
            rc = mi355x_gemv_fused(b->k, &d);
            if (rc != MI355X_E_UNSUPPORTED) { fused = true; rc_out = rc; end_out = ch.end; }
        }
    }
    if (!fused) {
        mi355x_tensor md = to_mt(fa);
        rc_out = mi355x_flash_attn_combine(b->k, &parts, &md);
    }
}

// ---------------------------------------------------------------------------------------------------
// supports_op / single-node dispatch
// ---------------------------------------------------------------------------------------------------
static bool whole_quant_ok(const ggml_tensor * t) {   // quantized operands must be whole contiguous tensors (planar layout)
    return !is_quant_type(t->type) || (ggml_is_contiguous(t) && (!t->view_src || (t->view_offs == 0 && ggml_nbytes(t) == ggml_nbytes(t->view_src))));
}

static bool mi_supports_op_impl(const ggml_tensor * op) {
    const ggml_tensor * s0 = op->src[0], * s1 = op->src[1];
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (op->type != GGML_TYPE_F32) return false;
            const ggml_type wt = s0->type;
            if (!(wt == GGML_TYPE_F32 || wt == GGML_TYPE_F16 || wt == GGML_TYPE_Q4_0 || wt == GGML_TYPE_Q5_0 || wt == GGML_TYPE_Q8_0 || wt == GGML_TYPE_Q4_K)) return false;
            if (s1->type != GGML_TYPE_F32 && s1->type != GGML_TYPE_F16) return false;
            // src1 strided along k (a transposed view: the voice-activity LSTM, src/whisper.cpp:4598-4602): the generic kernel only
            if (s1->nb[0] != ggml_type_size(s1->type)) return !is_quant_type(wt) && s0->nb[0] == ggml_type_size(wt) && s1->nb[0] % ggml_type_size(s1->type) == 0;
            if (is_quant_type(wt)) return whole_quant_ok(s0) && s0->ne[0] % 32 == 0;
            return s0->nb[0] == ggml_type_size(wt);
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            const ggml_tensor * k = s1, * v = op->src[2], * m = op->src[3];
            if (op->src[4]) return false;                                        // sinks
            float max_bias, softcap; memcpy(&max_bias, (const float *) op->op_params + 1, 4); memcpy(&softcap, (const float *) op->op_params + 2, 4);
            if (max_bias != 0.0f || softcap != 0.0f) return false;
            if (s0->type != GGML_TYPE_F32 || k->type != GGML_TYPE_F16 || v->type != GGML_TYPE_F16 || op->type != GGML_TYPE_F32) return false;
            if (s0->ne[0] != 64 || k->ne[0] != 64 || v->ne[0] != 64) return false;
            if (s0->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return false;
            if (s0->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2) return false;
            if ((s0->nb[1] | s0->nb[2] | k->nb[1] | k->nb[2] | v->nb[1] | v->nb[2]) % 16) return false;
            if (m && (m->type != GGML_TYPE_F16 || m->ne[2] != 1 || m->ne[3] != 1 || m->nb[0] != 2)) return false;
            return true;
        }
        case GGML_OP_ADD: case GGML_OP_MUL:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s1->type == GGML_TYPE_F32 && ggml_are_same_shape(op, s0);
        case GGML_OP_SCALE:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && ggml_is_contiguous(op) && ggml_is_contiguous(s0);
        case GGML_OP_UNARY: {
            // GELU: the whisper graphs; ReLU / sigmoid / tanh: the voice-activity-detection graph (src/whisper.cpp:4545-4680)
            const ggml_unary_op u = ggml_get_unary_op(op);
            return (u == GGML_UNARY_OP_GELU || u == GGML_UNARY_OP_RELU || u == GGML_UNARY_OP_SIGMOID || u == GGML_UNARY_OP_TANH) &&
                   op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && ggml_is_contiguous(op) && ggml_is_contiguous(s0);
        }
        case GGML_OP_SQRT:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && ggml_is_contiguous(op) && ggml_is_contiguous(s0);
        case GGML_OP_PAD_REFLECT_1D:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && op->op_params[0] < s0->ne[0] && op->op_params[1] < s0->ne[0];
        case GGML_OP_NORM:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s0->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP:
            return (s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_F16) && (op->type == GGML_TYPE_F32 || op->type == GGML_TYPE_F16);
        case GGML_OP_GET_ROWS:
            if (op->type != GGML_TYPE_F32 || s1->type != GGML_TYPE_I32) return false;
            if (is_quant_type(s0->type)) return whole_quant_ok(s0) && s0->ne[2] == 1 && s0->ne[3] == 1 && (op->nb[1] % 16 == 0);
            return s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_F16;
        case GGML_OP_IM2COL: {
            const bool is_2d = op->op_params[6] == 1;
            return !is_2d && s1->type == GGML_TYPE_F32 && s1->nb[0] == 4 && (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32) && s1->ne[3] == 1;
        }
        case GGML_OP_SOFT_MAX:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && !op->src[2] && s0->nb[0] == 4 &&
                   (!s1 || ((s1->type == GGML_TYPE_F32 || s1->type == GGML_TYPE_F16) && s1->ne[0] == s0->ne[0]));
        case GGML_OP_ROPE: {
            const int mode = op->op_params[2];
            if (mode == 24 && op->op_params[1] != s0->ne[0] / 2) return false;
            return (mode == 0 || mode == 2 || mode == 8 || mode == 24 || mode == 40) && op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s0->nb[0] == 4 && op->op_params[15] == 0;
        }
        case GGML_OP_CONCAT:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s1->type == GGML_TYPE_F32;
        default:
            return false;
    }
}

static int run_node(mi_backend_ctx * b, const ggml_tensor * n) {
    mi355x_ctx * k = b->k;
    switch (n->op) {
        case GGML_OP_ADD: case GGML_OP_MUL: {
            mi355x_tensor a = to_mt(n->src[0]), c = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_binary(k, n->op == GGML_OP_ADD ? 0 : 1, &a, &c, &d);
        }
        case GGML_OP_SCALE: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_scale(k, &a, &d, ggml_get_op_params_f32(n, 0), ggml_get_op_params_f32(n, 1));
        }
        case GGML_OP_UNARY: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            switch (ggml_get_unary_op(n)) {
                case GGML_UNARY_OP_GELU:    return mi355x_gelu(k, &a, &d);
                case GGML_UNARY_OP_RELU:    return mi355x_unary(k, MI355X_UNARY_RELU, &a, &d);
                case GGML_UNARY_OP_SIGMOID: return mi355x_unary(k, MI355X_UNARY_SIGMOID, &a, &d);
                case GGML_UNARY_OP_TANH:    return mi355x_unary(k, MI355X_UNARY_TANH, &a, &d);
                default: return MI355X_E_UNSUPPORTED;
            }
        }
        case GGML_OP_SQRT: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_unary(k, MI355X_UNARY_SQRT, &a, &d);
        }
        case GGML_OP_PAD_REFLECT_1D: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_pad_reflect_1d(k, &a, &d, n->op_params[0], n->op_params[1]);
        }
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_cpy(k, &a, &d);
        }
        case GGML_OP_GET_ROWS: {
            mi355x_tensor a = to_mt(n->src[0]), i = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_get_rows(k, &a, &i, &d);
        }
        case GGML_OP_IM2COL: {
            mi355x_tensor x = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_im2col_1d(k, &x, &d, (int) n->src[0]->ne[0], n->op_params[0], n->op_params[2], n->op_params[4]);
        }
        case GGML_OP_SOFT_MAX: {
            mi355x_tensor x = to_mt(n->src[0]), d = to_mt(n), m;
            if (n->src[1]) m = to_mt(n->src[1]);
            return mi355x_soft_max(k, &x, n->src[1] ? &m : nullptr, &d, ggml_get_op_params_f32(n, 0), ggml_get_op_params_f32(n, 1));
        }
        case GGML_OP_ROPE: {
            mi355x_rope_params p;
            p.n_dims = n->op_params[1]; p.mode = n->op_params[2]; p.n_ctx_orig = n->op_params[4];
            memcpy(&p.freq_base, n->op_params + 5, 4); memcpy(&p.freq_scale, n->op_params + 6, 4); memcpy(&p.ext_factor, n->op_params + 7, 4);
            memcpy(&p.attn_factor, n->op_params + 8, 4); memcpy(&p.beta_fast, n->op_params + 9, 4); memcpy(&p.beta_slow, n->op_params + 10, 4);
            memcpy(p.sections, n->op_params + 11, sizeof(int32_t) * 4);
            mi355x_tensor x = to_mt(n->src[0]), pos = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_rope(k, &x, &pos, n->src[2] ? (const float *) n->src[2]->data : nullptr, &d, &p);
        }
        case GGML_OP_CONCAT: {
            mi355x_tensor a = to_mt(n->src[0]), c = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_concat(k, &a, &c, &d, n->op_params[0]);
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            mi355x_tensor q = to_mt(n->src[0]), kk = to_mt(n->src[1]), v = to_mt(n->src[2]), d = to_mt(n), m;
            if (n->src[3]) m = to_mt(n->src[3]);
            float scale; memcpy(&scale, n->op_params, 4);
            return mi355x_flash_attn_ext(k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale);
        }
        default:
            return MI355X_E_UNSUPPORTED;
    }
}

// ---- the plane pipeline's stages (structures: mi_colset / mi_qstate above) -------------------------------------------
static const ggml_tensor * cs_tensor(const mi_colset & cs, int c, int node, int slot) {
    const ggml_tensor * n = (cs.S > 1 ? cs.g[c] : cs.g[0])->nodes[node];
    return slot < 0 ? n : n->src[slot];
}
// address of column c of the tensor at (node, src slot or -1); nb1 < 0: the tensor's own column stride
static char * cs_col(const mi_colset & cs, int c, int node, int slot, int64_t nb1 = -1) {
    const ggml_tensor * t = cs_tensor(cs, c, node, slot);
    if (cs.S > 1) return (char *) t->data;
    return (char *) t->data + (int64_t) c * (nb1 >= 0 ? nb1 : (int64_t) t->nb[1]);
}
static bool q_weight_ok(const ggml_tensor * w, int64_t K) {
    return is_quant_type(w->type) && ggml_is_contiguous(w) && ggml_n_dims(w) <= 2 && w->ne[0] == K && K % 32 == 0 && (w->type != GGML_TYPE_Q4_K || K % 256 == 0);
}
// segment s of a plane mat-vec from mul_mat chain `ch` (graph 0 describes the shapes, every column's graph its own addresses)
static void q_fill_seg(const mi_colset & cs, const mm_chain & ch, int s, mi355x_gemv_desc & d, mi355x_gemv_cols & cols) {
    const ggml_tensor * w = ch.mm->src[0];
    mi355x_gemv_seg & sg = d.seg[s];
    sg.w = w->data; sg.wtype = (int32_t) w->type; sg.N = (int32_t) w->ne[1]; sg.ep = ch.ep;
    sg.ep.residual = nullptr; sg.ep.residual_nb1 = 0;                      // per column, below
    sg.dst = nullptr; sg.dst_type = (int32_t) ch.last->type; sg.dst_nb1 = 0;
    const int64_t dst_nb1 = ch.last->type == GGML_TYPE_F16 ? (int64_t) w->ne[1]*2 : (ch.last == ch.mm ? (int64_t) ch.mm->nb[1] : (int64_t) ch.last->nb[1]);
    for (int c = 0; c < cs.T; c++) {
        cols.dst[s][c] = cs_col(cs, c, ch.end, -1, dst_nb1);
        cols.res[s][c] = ch.res_node >= 0 ? (const float *) cs_col(cs, c, ch.res_node, ch.res_slot) : nullptr;
    }
}

static bool q_mm(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, int & end_out, int & rc_out);

// norm -> mul -> add -> 1..3 mul_mat chains on the result (Q/K/V, cross-Q, fc1, logits).  k == nullptr: pattern check only.
static bool q_ln_gemv(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, const ln_chain & ln, int & end_out, int & rc_out) {
    const ggml_cgraph * g = cs.g[0];
    if (!ln.w || !ln.b) return false;
    const ggml_tensor * x = ln.norm->src[0], * lnout = ln.last;
    const int64_t K = x->ne[0];
    if (ggml_nrows(x) != (cs.S > 1 ? 1 : cs.T) || K > 2048 || K % 32 || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[0] != 4 || (x->nb[1] % 16) || ((uintptr_t) x->data % 16)) return false;
    if (((uintptr_t) ln.w % 16) || ((uintptr_t) ln.b % 16)) return false;
    const int nuse = use_count(g, lnout);
    if (nuse < 1 || nuse > 3 || lnout->view_src || (lnout->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    mm_chain ch[3];
    int j = next_real(g, ln.end), n = 0;
    while (n < nuse && j < g->n_nodes) {
        const ggml_tensor * t = g->nodes[j];
        if (t->op != GGML_OP_MUL_MAT || t->src[1] != lnout) return false;
        if (!parse_mm_chain(g, j, true, ch[n])) return false;
        j = next_real(g, ch[n].end);
        n++;
    }
    if (n != nuse) return false;
    for (int s = 0; s < n; s++) {
        const ggml_tensor * w = ch[s].mm->src[0];
        if (w->type != ch[0].mm->src[0]->type || !q_weight_ok(w, K)) return false;
        if (t_overlap(ch[s].last, x)) return false;
        for (int s2 = 0; s2 < s; s2++) if (t_overlap(ch[s].last, ch[s2].last)) return false;
    }
    // fc1 + GELU whose only reader is the next mat-vec: the epilogue writes that mat-vec's planes (and skips the F32 store if it can)
    bool pout = false, only = false;
    const ggml_tensor * w0 = ch[0].mm->src[0];
    if (n == 1 && w0->ne[1] % 32 == 0 && w0->ne[1] <= 8192 && w0->type != GGML_TYPE_Q4_K && ch[0].last->type == GGML_TYPE_F32 && ch[0].res_node < 0) {
        const int jn = next_real(g, ch[0].end);
        if (jn < g->n_nodes && g->nodes[jn]->op == GGML_OP_MUL_MAT && g->nodes[jn]->src[1] == ch[0].last) {
            // only if the consumer WILL take the planes (q_mm's own conditions): an elided F32 result exists nowhere else
            const ggml_tensor * w2 = g->nodes[jn]->src[0];
            mi_qstate dummy; int e2 = 0, r2 = 0;
            if (w2->type != GGML_TYPE_Q4_K && q_mm(nullptr, cs, dummy, jn, e2, r2)) { pout = true; only = can_elide(g, ch[0].last, 1); }
        }
    }
    end_out = ch[n - 1].end; rc_out = 0;
    if (!k) return true;
    // k_act_prepare -> planes -> mat-vec: two launches.  The LayerNorm in the mat-vec's own prologue (one launch) lost every time it was built: a
    // workgroup then normalises and quantizes ALL T columns (k_act_prepare spreads them over T workgroups) — k_gemv_q form r03: LN + Q/K/V 15.7 us
    // against 4.5 + 7.2; matrix-core form r05: 16 / 32 streams 10.6 / 7.8 chunks/s against 14.6 / 20.7, beam step 0.552 against 0.487 ms per token
    // (profiles/r05_ln_fused_ab.txt).  Both forms are deleted.
    mi355x_act_desc a; memset(&a, 0, sizeof(a));
    a.K = (int) K; a.T = cs.T; a.wtype = (int32_t) w0->type; a.has_norm = 1; memcpy(&a.eps, ln.norm->op_params, sizeof(float)); a.ln_w = ln.w; a.ln_b = ln.b;
    for (int c = 0; c < cs.T; c++) a.xcol[c] = (const float *) cs_col(cs, c, i, 0);
    void * p0 = mi355x_act_scratch(k, 0), * p1 = mi355x_act_scratch(k, 1);
    if (!p0 || !p1) { rc_out = (int) hipErrorOutOfMemory; return true; }
    int rc = 0;
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    mi355x_gemv_cols cols; memset(&cols, 0, sizeof(cols));
    d.K = (int) K; d.T = cs.T; d.nseg = n; d.cols = &cols;
    for (int s = 0; s < n; s++) q_fill_seg(cs, ch[s], s, d, cols);
    if (pout) { d.planes_out = p1; d.planes_out_only = only ? 1 : 0; }
    auto use_planes = [&]() -> int {
        const int r = mi355x_act_prepare(k, &a, p0);
        if (r) return r;
        d.x_planes = p0;
        return 0;
    };
    rc = use_planes();
    if (rc == MI355X_E_UNSUPPORTED) return false;
    if (rc) { rc_out = rc; return true; }
    bool mirror = n == 1 && cs.owner[0] && mirror_wanted(g, ch[0], cs.S > 1 ? 1 : cs.T);
    if (mirror) {
        const size_t rowb = (size_t) ch[0].last->ne[0] * 4;
        for (int c = 0; c < cs.T && mirror; c++) {
            mi_backend_ctx * ob = cs.S > 1 ? cs.owner[c] : cs.owner[0];
            char * md = ob ? mi_mirror_dev(ob) : nullptr;
            if (!md) mirror = false; else cols.mirror[c] = cs.S > 1 ? md : md + (size_t) c * rowb;
        }
        if (!mirror) memset(cols.mirror, 0, sizeof(cols.mirror));
    }
    rc = mi355x_gemv_fused(k, &d);
    if (rc == 0 && mirror && mi355x_last_launch_mirrored(k)) {
        const size_t rowb = (size_t) ch[0].last->ne[0] * 4;
        if (cs.S > 1) for (int c = 0; c < cs.T; c++) { mi_backend_ctx * ob = cs.owner[c]; ob->mirror_src = cs_tensor(cs, c, ch[0].end, -1)->data; ob->mirror_bytes = rowb; ob->mirror_state.store(1); }
        else { mi_backend_ctx * ob = cs.owner[0]; ob->mirror_src = ch[0].last->data; ob->mirror_bytes = rowb * (size_t) cs.T; ob->mirror_state.store(1); }
    }
    if (rc == MI355X_E_UNSUPPORTED && pout) { d.planes_out = nullptr; d.planes_out_only = 0; pout = false; rc = mi355x_gemv_fused(k, &d); }
    if (rc == MI355X_E_UNSUPPORTED) { rc_out = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: plane mat-vec rejected a shape its planes were already prepared for\n"); return true; }
    rc_out = rc;
    qs = mi_qstate();
    if (pout && rc == 0) { qs.src = ch[0].last->data; qs.K = w0->ne[1]; qs.T = cs.T; qs.which = 1; }
    return true;
}

// flash_attn_ext -> reshape -> mul_mat chain (output projection)
static bool q_attn_proj(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, int & end_out, int & rc_out) {
    const ggml_cgraph * g = cs.g[0];
    const ggml_tensor * fa = g->nodes[i];
    const ggml_tensor * q = fa->src[0], * kk = fa->src[1], * v = fa->src[2], * m = fa->src[3];
    const int64_t Tq = q->ne[1], H = q->ne[2];
    if (Tq != (cs.S > 1 ? 1 : cs.T) || q->ne[3] != 1 || fa->type != GGML_TYPE_F32 || !ggml_is_contiguous(fa) || H*64 > 2048) return false;
    const int j = next_real(g, i);
    if (!(j < g->n_nodes && g->nodes[j]->op == GGML_OP_MUL_MAT && can_elide(g, fa, 1))) return false;
    const ggml_tensor * mm = g->nodes[j], * x = mm->src[1], * w = mm->src[0];
    const bool x_is_fa = x->view_src == fa && x->view_offs == 0 && x->type == GGML_TYPE_F32 && ggml_is_contiguous(x) &&
                         x->ne[0] == H*64 && x->ne[1] == Tq && x->ne[2] == 1 && x->ne[3] == 1 && use_count(g, x) == 1 && !(x->flags & GGML_TENSOR_FLAG_OUTPUT);
    mm_chain ch;
    if (!x_is_fa || !q_weight_ok(w, H*64) || !parse_mm_chain(g, j, true, ch)) return false;
    end_out = ch.end; rc_out = 0;
    if (!k) return true;
    float scale; memcpy(&scale, fa->op_params, 4);
    mi355x_attn_partials parts;
    mi355x_tensor mq = to_mt(q), mk = to_mt(kk), mv = to_mt(v), mm_;
    int rc;
    void * p0 = mi355x_act_scratch(k, 0);
    if (!p0) { rc_out = (int) hipErrorOutOfMemory; return true; }
    // every column's operands (own graph in a cross-state batch; the token columns of the one graph otherwise)
    mi355x_attn_state st[MI355X_MAX_COLS]; memset(st, 0, sizeof(st));
    int max_kv = 0;
    for (int c = 0; c < cs.T; c++) {
        if (cs.S > 1) {
            const ggml_tensor * qc = cs_tensor(cs, c, i, 0), * kc = cs_tensor(cs, c, i, 1), * vc = cs_tensor(cs, c, i, 2), * mc = cs_tensor(cs, c, i, 3);
            if (kc->nb[1] != kk->nb[1] || kc->nb[2] != kk->nb[2] || vc->nb[1] != v->nb[1] || vc->nb[2] != v->nb[2] || qc->nb[2] != q->nb[2] || (mc != nullptr) != (m != nullptr) || vc->ne[1] != kc->ne[1]) {
                rc_out = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: cross-state batch: attention operands of the states are laid out differently\n"); return true;
            }
            st[c].q = qc->data; st[c].k = kc->data; st[c].v = vc->data; st[c].mask = mc ? mc->data : nullptr; st[c].n_kv = (int32_t) kc->ne[1];
        } else {
            st[c].q = (const char *) q->data + (int64_t) c * (int64_t) q->nb[1]; st[c].k = kk->data; st[c].v = v->data;
            st[c].mask = m ? (const char *) m->data + (int64_t) c * (int64_t) m->nb[1] : nullptr; st[c].n_kv = (int32_t) kk->ne[1];
        }
        max_kv = std::max(max_kv, (int) st[c].n_kv);
    }
    // ONE launch from q / K / V to the projection's activation planes for self-attention (<= 512 keys).  Cross-attention's 1500 keys the same
    // way (three rounds in one 16-wave workgroup per (head, column)) lost twice — r03: 13.5 us against 6.5 + 6.7 for partial records + combine;
    // r05 with the matrix-core mat-vecs: 16 / 32 streams 14.8 / 18.5 chunks/s against 15.7 / 21.9 (profiles/r05_stream_scaling.txt)
    constexpr int planes_max_kv = 512;
    bool have_planes = false;
    if (max_kv <= planes_max_kv && w->type != GGML_TYPE_Q4_K && (!m || (m->type == GGML_TYPE_F16 && m->nb[0] == 2))) {
        rc = mi355x_flash_attn_planes(k, cs.T, st, &mq, &mk, &mv, scale, p0);
        if (rc == 0) have_planes = true;
        else if (rc != MI355X_E_UNSUPPORTED) { rc_out = rc; return true; }
    }
    if (!have_planes) {
        if (cs.S > 1) rc = mi355x_flash_attn_partial_multi(k, cs.T, st, &mq, &mk, &mv, scale, &parts);
        else {
            if (m) mm_ = to_mt(m);
            rc = mi355x_flash_attn_partial(k, &mq, &mk, &mv, m ? &mm_ : nullptr, scale, &parts);
        }
        if (rc == MI355X_E_UNSUPPORTED) return false;
        if (rc) { rc_out = rc; return true; }
        mi355x_act_desc a; memset(&a, 0, sizeof(a));
        a.K = (int) (H*64); a.T = cs.T; a.wtype = (int32_t) w->type;
        a.attn_part_o = parts.part_o; a.attn_part_ml = parts.part_ml; a.attn_nparts = parts.nparts;
        rc = mi355x_act_prepare(k, &a, p0);
    }
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    mi355x_gemv_cols cols; memset(&cols, 0, sizeof(cols));
    d.K = (int) (H*64); d.T = cs.T; d.nseg = 1; d.x_planes = p0; d.cols = &cols;
    q_fill_seg(cs, ch, 0, d, cols);
    mi_fault_apply(g, i, d.seg[0].ep);
    if (rc == 0) rc = mi355x_gemv_fused(k, &d);
    if (rc == MI355X_E_UNSUPPORTED) { rc = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: plane pipeline rejected the attention output projection\n"); }
    rc_out = rc;
    qs = mi_qstate();
    return true;
}

// mul_mat chain on an F32 activation (fc2; any projection the patterns above did not take)
static bool q_mm(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, int & end_out, int & rc_out) {
    const ggml_cgraph * g = cs.g[0];
    mm_chain ch;
    if (!parse_mm_chain(g, i, true, ch)) return false;
    const ggml_tensor * w = ch.mm->src[0], * x = ch.mm->src[1];
    const int64_t K = w->ne[0];
    if (!q_weight_ok(w, K) || K > 8192 || x->type != GGML_TYPE_F32 || x->ne[1] != (cs.S > 1 ? 1 : cs.T) || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[0] != 4 ||
        (x->nb[1] % 16) || ((uintptr_t) x->data % 16) || ch.mm->type != GGML_TYPE_F32) return false;
    end_out = ch.end; rc_out = 0;
    if (!k) return true;
    void * planes;
    int rc = 0;
    if (qs.src == x->data && qs.K == K && qs.T == cs.T) planes = mi355x_act_scratch(k, qs.which);
    else {
        planes = mi355x_act_scratch(k, 0);
        mi355x_act_desc a; memset(&a, 0, sizeof(a));
        a.K = (int) K; a.T = cs.T; a.wtype = (int32_t) w->type;
        for (int c = 0; c < cs.T; c++) a.xcol[c] = (const float *) cs_col(cs, c, i, 1);
        rc = planes ? mi355x_act_prepare(k, &a, planes) : (int) hipErrorOutOfMemory;
        if (rc == MI355X_E_UNSUPPORTED) return false;
    }
    qs = mi_qstate();
    if (rc) { rc_out = rc; return true; }
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    mi355x_gemv_cols cols; memset(&cols, 0, sizeof(cols));
    d.K = (int) K; d.T = cs.T; d.nseg = 1; d.x_planes = planes; d.cols = &cols;
    q_fill_seg(cs, ch, 0, d, cols);
    rc = mi355x_gemv_fused(k, &d);
    if (rc == MI355X_E_UNSUPPORTED) { rc = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: plane mat-vec rejected a shape its planes were already prepared for\n"); }
    rc_out = rc;
    return true;
}

// One launch chain for S single-token decoder graphs (the columns).  k == nullptr: does every node fit?  (nothing is launched)
static int mi_walk_batch(mi355x_ctx * k, const mi_colset & cs) {
    const ggml_cgraph * g = cs.g[0];
    mi_qstate qs;
    static std::atomic<int> n_real_walks{0};
    const bool inject_reject = k && mi_fault().reject_chain > 0 && ++n_real_walks == mi_fault().reject_chain;      // (TEST fault injection, see mi_test_fault)
    for (int i = 0; i < g->n_nodes; i++) {
        const ggml_tensor * n = g->nodes[i];
        if (op_is_empty(n) || ggml_is_empty(n) || !(n->flags & GGML_TENSOR_FLAG_COMPUTE)) continue;
        if (inject_reject && i > g->n_nodes / 2) { mi355x_flush(k); GGML_LOG_WARN("ggml-mi355x: TEST fault: merged chain rejected at node %d of %d\n", i, g->n_nodes); return (int) hipErrorInvalidValue; }
        int end = i, rc = 0;
        bool took = false;
        if (n->op == GGML_OP_GET_ROWS) {
            // token embedding + positional embedding of every state in one launch
            const int j1 = next_real(g, i), j2 = j1 < g->n_nodes ? next_real(g, j1) : g->n_nodes;
            if (j2 < g->n_nodes && g->nodes[j1]->op == GGML_OP_GET_ROWS && g->nodes[j2]->op == GGML_OP_ADD) {
                const ggml_tensor * ga = n, * gb = g->nodes[j1], * ad = g->nodes[j2];
                const bool pair = (ad->src[0] == ga && ad->src[1] == gb) || (ad->src[0] == gb && ad->src[1] == ga);
                if (pair && can_elide(g, ga, 1) && can_elide(g, gb, 1) && ggml_are_same_shape(ga, gb) && ggml_are_same_shape(ad, ga) && ad->type == GGML_TYPE_F32 &&
                    ggml_is_contiguous(ad) && gb->src[0]->type == GGML_TYPE_F32 && ga->src[1]->type == GGML_TYPE_I32 && gb->src[1]->type == GGML_TYPE_I32 &&
                    ggml_nelements(ga->src[1]) == 1 && ggml_nelements(gb->src[1]) == 1 && whole_quant_ok(ga->src[0])) {
                    took = true; end = j2;
                    if (k) {
                        mi355x_head_state st[MI355X_MAX_COLS]; memset(st, 0, sizeof(st));
                        for (int c = 0; c < cs.T; c++) {
                            st[c].tok = (const int32_t *) cs_tensor(cs, c, i, 1)->data; st[c].pos = (const int32_t *) cs_tensor(cs, c, j1, 1)->data;
                            st[c].dst = (float *) cs_tensor(cs, c, j2, -1)->data;
                        }
                        mi355x_tensor te = to_mt(ga->src[0]), pe = to_mt(gb->src[0]);
                        rc = mi355x_decode_head_multi(k, cs.T, st, &te, &pe);
                    }
                }
            }
        } else if (n->op == GGML_OP_CPY || n->op == GGML_OP_CONT || n->op == GGML_OP_DUP) {
            // the mask row's F32 -> F16 cast (src/whisper.cpp:2520), every state's in one launch
            const ggml_tensor * s0 = n->src[0];
            if (s0->type == GGML_TYPE_F32 && n->type == GGML_TYPE_F16 && ggml_is_contiguous(s0) && ggml_is_contiguous(n) && ggml_nelements(s0) == ggml_nelements(n) &&
                ggml_nelements(n) < (1 << 20)) {
                took = true;
                if (k) {
                    mi355x_head_state st[MI355X_MAX_COLS]; memset(st, 0, sizeof(st));
                    for (int c = 0; c < cs.T; c++) {
                        const ggml_tensor * nc = cs_tensor(cs, c, i, -1);
                        st[c].mask_f32 = (const float *) nc->src[0]->data; st[c].mask_f16 = nc->data; st[c].n_mask = (int32_t) ggml_nelements(nc);
                    }
                    rc = mi355x_decode_head_multi(k, cs.T, st, nullptr, nullptr);          // no state embeds: the tables are not needed
                }
            }
        } else if (n->op == GGML_OP_NORM) {
            ln_chain c;
            parse_ln_chain(g, i, true, c);
            took = q_ln_gemv(k, cs, qs, i, c, end, rc);
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT) {
            took = q_attn_proj(k, cs, qs, i, end, rc);
        } else if (n->op == GGML_OP_MUL_MAT) {
            took = q_mm(k, cs, qs, i, end, rc);
        }
        if (!took) return MI355X_E_UNSUPPORTED;
        if (rc != 0) { GGML_LOG_ERROR("ggml-mi355x: cross-state batch: op %s (%s) failed: rc=%d %s\n", ggml_op_name(n->op), n->name, rc, mi355x_last_error()); return rc; }
        i = end;
    }
    return k ? mi355x_flush(k) : 0;
}

// may graph `b` run as another column next to graph `a`?  Same node sequence, shapes (up to the key counts) and weights.
static bool mi_graphs_congruent(const ggml_cgraph * a, const ggml_cgraph * b) {
    if (a->n_nodes != b->n_nodes) return false;
    for (int i = 0; i < a->n_nodes; i++) {
        const ggml_tensor * x = a->nodes[i], * y = b->nodes[i];
        // (extents are not compared: the key count n_kv — mask rows, K / V views — legitimately differs between states that are at
        //  different positions; with equal ops, types and WEIGHTS every other extent follows from the model)
        if (x->op != y->op || x->type != y->type || (x->flags & GGML_TENSOR_FLAG_COMPUTE) != (y->flags & GGML_TENSOR_FLAG_COMPUTE)) return false;
        for (int s = 0; s < 4; s++) {
            const ggml_tensor * xs = x->src[s], * ys = y->src[s];
            if ((xs == nullptr) != (ys == nullptr)) return false;
            if (!xs) continue;
            if (xs->type != ys->type) return false;
            ggml_backend_buffer_t xb = xs->view_src ? xs->view_src->buffer : xs->buffer;
            if (xb && xb->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS && xs->data != ys->data) return false;     // the same weights
        }
    }
    return true;
}

// walk nodes [i0, i_stop) and emit kernels on the backend's stream
static int mi_emit_range(mi_backend_ctx * b, ggml_cgraph * g, int i0, int i_stop) {
    int i = i0;
    const uint64_t trace_n0 = g_trace ? mi355x_eager_count(b->k) : 0;
    bool trace_first = g_trace && b->trace_gc_enter != 0;
    for (; i < i_stop; i++) {
        if (trace_first && mi355x_eager_count(b->k) != trace_n0) { g_trace_ns[2] += trace_now() - b->trace_gc_enter; g_trace_calls[2]++; trace_first = false; }
        const ggml_tensor * n = g->nodes[i];
        if (op_is_empty(n) || ggml_is_empty(n) || !(n->flags & GGML_TENSOR_FLAG_COMPUTE)) continue;
        int rc = MI355X_E_UNSUPPORTED;
        // decoder steps with planes_min_t .. 8 columns (beam search): the pre-quantized-activation pipeline (stages above); whatever it
        // does not take falls through to the fused / generic paths below
        constexpr int planes_min_t = 3;
        if (b->fuse && !b->exact && (n->op == GGML_OP_MUL_MAT || n->op == GGML_OP_NORM || n->op == GGML_OP_FLASH_ATTN_EXT)) {
            const int64_t Tn = n->op == GGML_OP_FLASH_ATTN_EXT ? n->src[0]->ne[1] : (n->op == GGML_OP_MUL_MAT ? n->src[1]->ne[1] : ggml_nrows(n->src[0]));
            if (Tn >= planes_min_t && Tn <= MI355X_IMG_COLS) {
                mi_colset cs; cs.S = 1; cs.T = (int) Tn; cs.g[0] = g; cs.owner[0] = b;
                int end = i, rc2 = 0; bool took = false;
                if (n->op == GGML_OP_NORM) { ln_chain c; parse_ln_chain(g, i, true, c); took = q_ln_gemv(b->k, cs, b->qs, i, c, end, rc2); }
                else if (n->op == GGML_OP_FLASH_ATTN_EXT) took = q_attn_proj(b->k, cs, b->qs, i, end, rc2);
                else took = q_mm(b->k, cs, b->qs, i, end, rc2);
                if (took) {
                    if (rc2 != 0) { GGML_LOG_ERROR("ggml-mi355x: op %s (%s) failed in the plane pipeline: rc=%d %s\n", ggml_op_name(n->op), n->name, rc2, mi355x_last_error()); return rc2; }
                    b->act_src = nullptr;
                    i = end;
                    continue;
                }
            }
        }
        b->qs = mi_qstate();
        if (n->op == GGML_OP_MUL_MAT) {
            mm_chain c;
            parse_mm_chain(g, i, b->fuse, c);
            rc = run_mm_chain(b, c, g);
            if (rc == MI355X_E_UNSUPPORTED && c.end != i) { parse_mm_chain(g, i, false, c); rc = run_mm_chain(b, c); }
            else i = c.end;
        } else if (n->op == GGML_OP_NORM) {
            ln_chain c;
            parse_ln_chain(g, i, b->fuse, c);
            int end = 0, rc2 = 0;
            if (b->fuse && try_ln_gemv(b, g, c, end, rc2)) { rc = rc2; i = end; }
            else {
                rc = run_ln_chain(b, c, g);                  // leaves b->act_* describing the prepared activations, if it made them
                if (rc == MI355X_E_UNSUPPORTED && c.end != i) { parse_ln_chain(g, i, false, c); rc = run_ln_chain(b, c); }
                else i = c.end;
            }
        } else if (n->op == GGML_OP_GET_ROWS && b->fuse) {
            // token embedding + positional embedding: get_rows, get_rows, add -> one launch
            rc = MI355X_E_UNSUPPORTED;
            const int j1 = next_real(g, i), j2 = j1 < g->n_nodes ? next_real(g, j1) : g->n_nodes;
            if (j2 < g->n_nodes && g->nodes[j1]->op == GGML_OP_GET_ROWS && g->nodes[j2]->op == GGML_OP_ADD) {
                const ggml_tensor * ga = n, * gb = g->nodes[j1], * ad = g->nodes[j2];
                const bool pair = (ad->src[0] == ga && ad->src[1] == gb) || (ad->src[0] == gb && ad->src[1] == ga);
                if (pair && can_elide(g, ga, 1) && can_elide(g, gb, 1) && ggml_are_same_shape(ga, gb) && ggml_are_same_shape(ad, ga) &&
                    ad->type == GGML_TYPE_F32 && ggml_is_contiguous(ad) && gb->src[0]->type == GGML_TYPE_F32) {
                    mi355x_tensor sa = to_mt(ga->src[0]), ia = to_mt(ga->src[1]), sb = to_mt(gb->src[0]), ib = to_mt(gb->src[1]), d = to_mt(ad);
                    rc = mi355x_get_rows_add(b->k, &sa, &ia, &sb, &ib, &d);
                    if (rc != MI355X_E_UNSUPPORTED) i = j2;
                }
            }
            if (rc == MI355X_E_UNSUPPORTED) rc = run_node(b, n);
            b->act_src = nullptr;
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT && b->exact) {
            mi355x_tensor q = to_mt(n->src[0]), kk = to_mt(n->src[1]), v = to_mt(n->src[2]), d = to_mt(n), m;
            if (n->src[3]) m = to_mt(n->src[3]);
            float scale; memcpy(&scale, n->op_params, 4);
            rc = mi355x_flash_attn_ext_exact(b->k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale, b->n_threads);
            if (rc == MI355X_E_UNSUPPORTED) rc = run_node(b, n);
            b->act_src = nullptr;
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT && b->fuse && n->src[0]->ne[1] > 8) {
            // encoder / prompt attention whose result (through a reshape) is the activation matrix of the output projection: the
            // attention kernel leaves that GEMM's prepared f16 activations as well (one launch and one pass over the result less)
            constexpr bool on = true;
            const int j = next_real(g, i);
            const int64_t T = n->src[0]->ne[1], NS = n->ne[0] * n->ne[1];
            int mode = -1;
            rc = MI355X_E_UNSUPPORTED;
            if (on && j < g->n_nodes && g->nodes[j]->op == GGML_OP_MUL_MAT) {
                const ggml_tensor * x = g->nodes[j]->src[1];
                if (x->data == n->data && x->ne[0] == NS && x->ne[1] == T && (int64_t) x->nb[1] == NS*4 && ggml_is_contiguous(n) &&
                    mm_takes_prepared(b, g->nodes[j], x, mode) && (mode == 1 || mode == 3) && mi_act_reserve(b, (size_t) T * NS * 2) == 0) {
                    mi355x_tensor q = to_mt(n->src[0]), kk = to_mt(n->src[1]), v = to_mt(n->src[2]), d = to_mt(n), m;
                    if (n->src[3]) m = to_mt(n->src[3]);
                    float scale; memcpy(&scale, n->op_params, 4);
                    rc = mode == 3 ? mi355x_flash_attn_ext_prep_rows(b->k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale, b->act)
                                   : mi355x_flash_attn_ext_prep(b->k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale, b->act);
                    if (rc == 0) { b->act_src = x->data; b->act_K = NS; b->act_T = T; b->act_mode = mode; b->act_nb1 = NS*4; }
                }
            }
            if (rc == MI355X_E_UNSUPPORTED) { rc = run_node(b, n); b->act_src = nullptr; }
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT && b->fuse && n->src[0]->ne[1] <= 8) {
            int end = i, rc2 = MI355X_E_UNSUPPORTED;
            if (try_fattn_gemv(b, g, i, end, rc2)) { rc = rc2; i = end; }
            else rc = run_node(b, n);
            b->act_src = nullptr;
        } else {
            rc = run_node(b, n);
            b->act_src = nullptr;
        }
        if (rc != 0) {
            GGML_LOG_ERROR("ggml-mi355x: op %s (%s) failed: rc=%d %s\n", ggml_op_name(n->op), n->name, rc, mi355x_last_error());
            return rc;
        }
    }
    return mi355x_flush(b->k);            // launches the kernel library held back for grouping (gemm_mfma.hip) leave with their range
}
static int mi_emit_graph(mi_backend_ctx * b, ggml_cgraph * g) {
    b->act_src = nullptr; b->elided_src = nullptr; b->elided_for = nullptr; b->qs = mi_qstate();
    return mi_emit_range(b, g, 0, g->n_nodes);
}

static const char * mi_backend_get_name(ggml_backend_t backend) { return ((mi_backend_ctx *) backend->context)->name.c_str(); }

static void mi_batch_leave(mi_backend_ctx * b);

static void mi_backend_free(ggml_backend_t backend) {
    mi_backend_ctx * b = (mi_backend_ctx *) backend->context;
    (void) hipSetDevice(b->device);
    mi_batch_leave(b);
    if (t_last_backend == b) t_last_backend = nullptr;
    mi355x_ctx_synchronize(b->k);
    if (b->batch_wait_sync) (void) hipEventSynchronize(b->batch_wait_sync);
    if (b->own_ev) (void) hipEventDestroy(b->own_ev);
    b->mirror_state.store(0);
    if (g_debug()) fprintf(stderr, "ggml-mi355x: backend %s: graph_compute=%" PRIu64 "\n", b->name.c_str(), b->n_graph_compute);
    if (b->span_pending) mi_span_drain(b);
    for (auto & e : b->span_ev) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
    if (b->act) (void) hipFree(b->act);
    if (b->act_alt) (void) hipFree(b->act_alt);
    {
        std::lock_guard<std::mutex> lk(g_weights_mtx);
        std::unique_lock<std::shared_mutex> wl(g_backends_rw);
        for (size_t i = 0; i < g_backends.size(); i++) if (g_backends[i] == b) { g_backends.erase(g_backends.begin() + i); break; }
        g_total_stats[0] += b->n_graph_compute;
        g_total_host_ms[3] += b->t_eager_ms;
        g_total_gpu_span_ms += b->t_gpu_span_ms;
    }
    mi355x_ctx_destroy(b->k);
    if (b->mirror_host) (void) hipHostFree(b->mirror_host);
    delete b;
    delete backend;
}

static void mi_backend_synchronize(ggml_backend_t backend) {
    io_timer tm(3);
    mi_backend_ctx * b = (mi_backend_ctx *) backend->context;
    (void) hipSetDevice(b->device);
    mi355x_ctx_synchronize(b->k);
    if (b->batch_wait_sync) { (void) hipEventSynchronize(b->batch_wait_sync); b->batch_wait_sync = nullptr; }      // the launch chain this state was a column of
    { int one = 1; b->mirror_state.compare_exchange_strong(one, 2); }                                              // the mirrored logits have landed in host memory
    if (g_trace && b->trace_gc_enter) { g_trace_ns[4] += trace_now() - b->trace_gc_enter; g_trace_calls[4]++; b->trace_gc_enter = 0; }
    if (b->span_pending) mi_span_drain(b);
}

// one graph on the backend's own stream
static ggml_status mi_compute_own(mi_backend_ctx * b, ggml_cgraph * cgraph) {
    hipStream_t cs = (hipStream_t) mi355x_ctx_stream(b->k);
    if (b->batch_wait_stream) { (void) hipStreamWaitEvent(cs, b->batch_wait_stream, 0); b->batch_wait_stream = nullptr; }    // a batch wrote this state's KV / activations
    b->own_dirty = true;
    mi_io_order_stream(b->device, b->io, cs);
    // GPU span bookkeeping (two event records per call)
    constexpr bool span_on = true;
    int span_idx = -1;
    if (span_on && !b->prof) {
        if (b->span_ev.empty()) {
            b->span_ev.resize(64);
            for (auto & e : b->span_ev) { (void) hipEventCreate(&e.first); (void) hipEventCreate(&e.second); }
        }
        if (b->span_pending >= (int) b->span_ev.size()) { mi355x_ctx_synchronize(b->k); mi_span_drain(b); }
        span_idx = b->span_next; b->span_next = (b->span_next + 1) % (int) b->span_ev.size(); b->span_pending++;
        (void) hipEventRecord(b->span_ev[span_idx].first, cs);
    }
    struct span_end { mi_backend_ctx * b; int idx; ~span_end() { if (idx >= 0) (void) hipEventRecord(b->span_ev[idx].second, (hipStream_t) mi355x_ctx_stream(b->k)); } } span_guard{ b, span_idx };
    const double t0 = now_ms();
    const int rc = mi_emit_graph(b, cgraph);
    b->t_eager_ms += now_ms() - t0;
    return rc == 0 ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
}

// ---------------------------------------------------------------------------------------------------
// cross-state batches: the rendezvous.  Every whisper_state has its own ggml_backend_t (own host thread, own HIP stream: src/
// whisper.cpp:7848-7869).  When batching is on (the default; GGML_MI355X_BATCH=0 / ggml_backend_mi355x_set_batching(0) switch it off), a backend whose graph is a
// single-token decoder step does not launch it: it joins its device's group, and once every backend that is currently decoding has
// arrived (or the window closes) ONE of the waiting threads launches the merged chain on the group's stream (mi_walk_batch) — up to
// MI355X_MAX_COLS states as the columns of one pass over the weights.  A state that stops decoding (its next graph is an encoder,
// a prompt, a beam-search step) leaves the group at once, so nobody waits for it; one that simply stays away is dropped after the
// window.  Stream order: the group's stream waits for each member's earlier work on its own stream (encoder -> cross-KV), each
// member's stream and synchronize() wait for the batch's completion event.  One state alone runs exactly the non-batched path.
// ---------------------------------------------------------------------------------------------------
#define MI_BATCH_LANES 4
struct mi_batch_member { mi_backend_ctx * b; ggml_cgraph * g; int state; ggml_status status; };       // state: 0 waiting, 1 being launched, 2 done
struct mi_batch_group {
    std::mutex m; std::condition_variable cv;
    std::vector<mi_batch_member *> waiting;
    std::vector<mi_backend_ctx *>  members;          // backends currently counted in n_active
    double      last_finish_ms = 0;
    // up to MI_BATCH_LANES merged chains in flight at once, each on its own stream (own scratch arena, activation planes, event ring):
    // with more decoding states than columns per chain (GGML_MI355X_BATCH_COLS) the chains of different state groups overlap on the GPU
    struct lane { mi355x_ctx * k = nullptr; mi_io_marks io; hipEvent_t ev_ring[16] = {}; int ev_next = 0; bool busy = false; } lanes[MI_BATCH_LANES];
    int         lane_cols[MI_BATCH_LANES] = {};      // columns of the chain each busy lane is launching
    std::mutex  sig_m;
    uint64_t    sig_nodes = 0; const void * sig_w = nullptr;   // graph shape the dry walk + congruence check last accepted (sig_m)
    uint64_t    n_batches = 0, n_columns = 0, n_solo = 0, n_timeouts = 0;
    std::atomic<uint64_t> n_fallback{0};                 // (updated outside the group lock)
};
static mi_batch_group     g_batch[MI_MAX_DEVICES];
static std::atomic<int>   g_batching{-1};            // -1: not decided yet (environment), 0 off, 1 on (from mi_batch_min_states() states), n >= 2: on from n states
static bool mi_batching_on() {
    int v = g_batching.load();
    // on by default (r04): fewer than mi_batch_min_states() decoding states keep their own chains anyway, and beyond four states own
    // chains collapse (8 states: 1.8 chunks/s against 9.4 merged) — a whisper_full_parallel user must not have to know a switch
    if (v < 0) { const char * e = getenv("GGML_MI355X_BATCH"); v = e ? std::max(0, atoi(e)) : 1; g_batching.store(v); }
    return v != 0;
}
// Fewer decoding states than this run their own chains side by side (states-on-streams) although batching is on: a merged chain costs
// 12 launches per layer against 8 and moves the states in lockstep (their host phases no longer hide behind each other's GPU work) —
// measured large-v3 Q5_0: 2 / 4 states 3.35 / 5.96 chunks/s merged against 4.1 / 7.5 on their own streams, 8 states 9.75 against 2.8
// (profiles/r03_stream_scaling_*).  ggml_backend_mi355x_set_batching(n >= 2) / GGML_MI355X_BATCH=n sets the threshold to n.
static int mi_batch_min_states() {
    constexpr int env_min = 5;
    const int v = g_batching.load();
    return v >= 2 ? v : env_min;
}

// a single-token decoder step?  (cheap signature; whether every node fits is decided once per graph shape by the dry walk)
static bool mi_is_step_graph(const ggml_cgraph * g) {
    if (g->n_nodes < 32) return false;
    const ggml_tensor * last = g->nodes[g->n_nodes - 1];
    if (last->op != GGML_OP_MUL_MAT || last->ne[1] != 1 || last->ne[2] != 1 || last->ne[3] != 1) return false;
    for (int i = 0; i < g->n_nodes && i < 16; i++) {
        const ggml_tensor * n = g->nodes[i];
        if (op_is_empty(n)) continue;
        return n->op == GGML_OP_GET_ROWS && ggml_nelements(n->src[1]) == 1;
    }
    return false;
}

static void mi_batch_leave(mi_backend_ctx * b) {
    if (!b->in_group) return;
    mi_batch_group & grp = g_batch[b->device];
    std::lock_guard<std::mutex> lk(grp.m);
    if (!b->in_group) return;
    b->in_group = false;
    for (size_t i = 0; i < grp.members.size(); i++) if (grp.members[i] == b) { grp.members.erase(grp.members.begin() + i); break; }
    grp.cv.notify_all();
}

static ggml_status mi_compute_own(mi_backend_ctx * b, ggml_cgraph * cgraph);

// the merged launch chain for `n` members (group lock NOT held).  Falls back to every member alone when the graphs do not fit.
// returns true when the members left as ONE merged chain
static bool mi_compute_batch(mi_batch_group & grp, mi_batch_group::lane & ln, mi_batch_member ** mem, int n) {
    mi_backend_ctx * b0 = mem[0]->b;
    (void) hipSetDevice(b0->device);
    bool ok = true;
    if (!ln.k) {
        ln.k = mi355x_ctx_create(b0->device);
        for (auto & e : ln.ev_ring) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) ok = false;
        if (!ln.k) ok = false;
    }
    mi_colset cs; cs.S = n; cs.T = n;
    for (int c = 0; c < n; c++) { cs.g[c] = mem[c]->g; cs.owner[c] = mem[c]->b; }
    if (ok) {
        const ggml_cgraph * g0 = cs.g[0];
        const uint64_t sn = (uint64_t) g0->n_nodes; const void * sw = g0->nodes[g0->n_nodes - 1]->src[0]->data;
        bool same = true;
        for (int c = 1; c < n; c++) same = same && cs.g[c]->n_nodes == g0->n_nodes && cs.g[c]->nodes[g0->n_nodes - 1]->src[0]->data == sw;
        std::lock_guard<std::mutex> sl(grp.sig_m);
        if (!same) ok = false;
        else {
            // every state is checked ONCE per graph shape — node for node against the chain's first graph — before it may be a column,
            // not only the states that happened to be in the first batch of that shape
            bool fresh = grp.sig_nodes != sn || grp.sig_w != sw;
            // a state already verified for this shape (an unverified state in column 0 used to be compared with itself only)
            int vref = -1;
            for (int c = 0; c < n && vref < 0; c++) if (mem[c]->b->sig_nodes == sn && mem[c]->b->sig_w == sw) vref = c;
            if (vref < 0) fresh = true;                                      // nobody here has been checked: the dry walk below vouches for column 0
            for (int c = 0; c < n && ok; c++) {
                mi_backend_ctx * bc = mem[c]->b;
                if (!fresh && bc->sig_nodes == sn && bc->sig_w == sw) continue;          // verified earlier, and so is the graph it is compared with (transitively)
                const int ref = fresh ? 0 : vref;                           // fresh: everybody against column 0 (walked below); else against a verified member
                if (c != ref) ok = mi_graphs_congruent(cs.g[ref], cs.g[c]);
            }
            if (ok && fresh) ok = mi_walk_batch(nullptr, cs) == 0;
            if (ok) { grp.sig_nodes = sn; grp.sig_w = sw; for (int c = 0; c < n; c++) { mem[c]->b->sig_nodes = sn; mem[c]->b->sig_w = sw; } }
            else for (int c = 0; c < n; c++) mem[c]->b->no_batch_nodes = g0->n_nodes;       // this graph shape never batches: stop joining with it
        }
    }
    if (!ok) {
        grp.n_fallback++;
        for (int c = 0; c < n; c++) mem[c]->status = mi_compute_own(mem[c]->b, mem[c]->g);
        return false;
    }
    hipStream_t bs = (hipStream_t) mi355x_ctx_stream(ln.k);
    for (int c = 0; c < n; c++) {
        mi_backend_ctx * b = mem[c]->b;
        if (b->own_dirty) {                                   // the member's earlier work on its own stream (encoder -> cross-KV, a solo step's KV writes)
            (void) mi355x_flush(b->k);
            if (!b->own_ev) (void) hipEventCreateWithFlags(&b->own_ev, hipEventDisableTiming);
            (void) hipEventRecord(b->own_ev, (hipStream_t) mi355x_ctx_stream(b->k));
            (void) hipStreamWaitEvent(bs, b->own_ev, 0);
            b->own_dirty = false;
        }
        // the previous chain this state was a column of may have run on ANOTHER lane: its KV-cache and activation writes must be
        // ordered in front of this chain by the streams themselves, not by the host synchronize whisper happens to call between steps
        // (graph_compute is an asynchronous entry point)
        if (b->batch_wait_stream) (void) hipStreamWaitEvent(bs, b->batch_wait_stream, 0);
    }
    mi_io_order_stream(b0->device, ln.io, bs);                // every member's graph inputs leave with one scatter launch at the head of the chain
    const int rc = mi_walk_batch(ln.k, cs);
    if (rc != 0) {
        (void) mi355x_ctx_synchronize(ln.k);
        grp.n_fallback++;
        if (rc != MI355X_E_UNSUPPORTED && rc != (int) hipErrorInvalidValue) {
            // a device fault, not a rejection: repeating the step n times on the states' own chains would hit the same fault n times
            GGML_LOG_ERROR("ggml-mi355x: cross-state batch failed mid-chain (rc=%d %s): %d states report failure\n", rc, mi355x_last_error(), n);
            for (int c = 0; c < n; c++) mem[c]->status = GGML_STATUS_FAILED;
            return false;
        }
        // a kernel rejected the chain half-way (a shape or alignment only its launch code knows): what was launched has written nothing a
        // repeat would not write again (activations, this position's KV rows), so every member runs its step again on its own chain
        // once the partial chain has drained, and this graph shape stops batching for these states
        GGML_LOG_WARN("ggml-mi355x: cross-state batch rejected mid-chain (rc=%d %s): %d states repeat the step on their own chains\n", rc, mi355x_last_error(), n);
        { std::lock_guard<std::mutex> sl(grp.sig_m); grp.sig_nodes = 0; grp.sig_w = nullptr; }
        for (int c = 0; c < n; c++) {
            mem[c]->b->no_batch_nodes = mem[c]->g->n_nodes; mem[c]->b->sig_nodes = 0;
            mem[c]->status = mi_compute_own(mem[c]->b, mem[c]->g);
        }
        return false;
    }
    hipEvent_t ev = ln.ev_ring[ln.ev_next]; ln.ev_next = (ln.ev_next + 1) % 16;
    (void) hipEventRecord(ev, bs);
    for (int c = 0; c < n; c++) {
        mem[c]->b->batch_wait_sync = ev; mem[c]->b->batch_wait_stream = ev;
        mem[c]->status = GGML_STATUS_SUCCESS;
    }
    return true;
}

static ggml_status mi_batch_join(mi_backend_ctx * b, ggml_cgraph * cgraph) {
    // (the window must stay above a chain step: 1 ms / 0.4 ms collapse to 2.4 / 2.0 chunks/s at 16 streams — states are dropped while they are simply on
    //  their way through the host part of a step; profiles/r05_stream_scaling.txt)
    constexpr double window_ms = 3.0;
    // columns per merged chain.  GGML_MI355X_BATCH_COLS=n (2..32) fixes it; by default 60 % of the decoding states (at least 4) ride one chain and
    // the rest a second one next to it (MI_BATCH_LANES streams): two chains of unequal width fill each other's launch gaps.  Measured on large-v3
    // Q5_0 (profiles/r04_stream_scaling.txt, r04_chain_split_sweep.txt): 16 states as 10 + 6: 14.2 chunks/s, 12 + 4: 13.9, 8 + 8: 11.5-12.5, one
    // chain of 16: 12.9; 32 as 20 + 12: 18.0, 16 + 16: 15.3; 8 as 5 + 3 or 6 + 2: 9.4-10.0, one chain of 8: 9.3; 6 as 4 + 2: 8.2, one chain of 6: 7.4; 7 as 5 + 2: 8.7, 6 + 1: 7.2; three or more chains
    // (40 %): 11.0 at 32; chains of 3 + 2 at 5 states: 4.9 (one chain of 4 + a solo state: 6.6).
    // More than 8 columns travel as images of 8 (mi355x_kernels.h: MI355X_IMG_COLS): the weights are still read once per chain step.
    static const int env_cols = getenv("GGML_MI355X_BATCH_COLS") ? std::max(2, std::min(MI355X_MAX_COLS, atoi(getenv("GGML_MI355X_BATCH_COLS")))) : 0;
    mi_batch_group & grp = g_batch[b->device];
    mi_batch_member me = { b, cgraph, 0, GGML_STATUS_SUCCESS };
    std::unique_lock<std::mutex> lk(grp.m);
    if (!b->in_group) { b->in_group = true; grp.members.push_back(b); }
    constexpr int split_pct = 60, split_min = 4;
    auto cols_cap = [&]() {
        if (env_cols) return env_cols;
        return std::min(MI355X_MAX_COLS, std::max(split_min, (split_pct * (int) grp.members.size() + 99) / 100));
    };
    if ((int) grp.members.size() < mi_batch_min_states()) {
        // too few decoding states for a merged chain to pay: this step runs on the state's own stream (it stays counted)
        bool idle = true;
        for (int i = 0; i < MI_BATCH_LANES; i++) idle = idle && !grp.lanes[i].busy;
        if (idle && grp.waiting.empty()) { grp.n_solo++; lk.unlock(); return mi_compute_own(b, cgraph); }
    }
    grp.waiting.push_back(&me);
    const double arrived = now_ms();
    grp.cv.notify_all();                                      // a waiter may now have its full set
    for (;;) {
        if (me.state == 2) return me.status;
        int lane = -1;
        for (int i = 0; i < MI_BATCH_LANES && lane < 0; i++) if (!grp.lanes[i].busy) lane = i;
        bool lead = false;
        if (me.state == 0 && lane >= 0) {
            // states that are on their way through a running chain come back later: the set to wait for is everybody else
            int in_flight = 0;
            for (int i = 0; i < MI_BATCH_LANES; i++) if (grp.lanes[i].busy) in_flight += grp.lane_cols[i];
            const int want = std::max(1, std::min<int>((int) grp.members.size() - in_flight, cols_cap()));
            if ((int) grp.waiting.size() >= want) lead = true;
            else if (grp.waiting.front() == &me && now_ms() > std::max(arrived, grp.last_finish_ms) + window_ms) {
                // the window closed: whoever is counted but neither here nor a column of a running chain is dropped (it rejoins with its
                // next step)
                for (size_t i = 0; i < grp.members.size(); ) {
                    bool here = grp.members[i]->in_flight;
                    for (auto * w : grp.waiting) here = here || w->b == grp.members[i];
                    if (!here) { grp.members[i]->in_group = false; grp.members.erase(grp.members.begin() + i); } else i++;
                }
                grp.n_timeouts++;
                lead = true;
            }
        }
        if (!lead) {
            grp.cv.wait_for(lk, std::chrono::microseconds(200));
            continue;
        }
        mi_batch_member * mem[MI355X_MAX_COLS];
        int n = 0;
        const int max_cols = cols_cap();
        while (n < max_cols && !grp.waiting.empty()) { mem[n] = grp.waiting.front(); mem[n]->state = 1; mem[n]->b->in_flight = true; grp.waiting.erase(grp.waiting.begin()); n++; }
        mi_batch_group::lane & ln = grp.lanes[lane];
        ln.busy = true; grp.lane_cols[lane] = n;
        lk.unlock();
        bool merged = false;
        if (n == 1) { mem[0]->status = mi_compute_own(mem[0]->b, mem[0]->g); }
        else        merged = mi_compute_batch(grp, ln, mem, n);
        lk.lock();
        if (n == 1) grp.n_solo++; else if (merged) { grp.n_batches++; grp.n_columns += (uint64_t) n; }
        for (int c = 0; c < n; c++) { mem[c]->state = 2; mem[c]->b->in_flight = false; }
        ln.busy = false; grp.lane_cols[lane] = 0;
        grp.last_finish_ms = now_ms();
        grp.cv.notify_all();
    }
}

static ggml_status mi_backend_graph_compute(ggml_backend_t backend, ggml_cgraph * cgraph) {
    mi_backend_ctx * b = (mi_backend_ctx *) backend->context;
    if (hipSetDevice(b->device) != hipSuccess) return GGML_STATUS_FAILED;
    b->n_graph_compute++;
    trace_scope trace_gc(3);
    if (g_trace) b->trace_gc_enter = trace_gc.t0;
    b->mirror_state.store(0);                   // the new graph reuses the compute buffer the mirrored tensor lived in
    t_last_backend = b;
    {   // the logits of a decoder graph: its last node, a mat-vec / mat-mul over the vocabulary (src/whisper.cpp:2827)
        const ggml_tensor * last = cgraph->n_nodes > 0 ? cgraph->nodes[cgraph->n_nodes - 1] : nullptr;
        if (last && last->op == GGML_OP_MUL_MAT && last->type == GGML_TYPE_F32 && last->ne[0] > 8192 && (int64_t) last->nb[1] == last->ne[0]*4 && last->ne[2] == 1 && last->ne[3] == 1) {
            b->logits_dev = (const float *) last->data; b->logits_n = (int) last->ne[0]; b->logits_rows = (int) last->ne[1];
        } else b->logits_dev = nullptr;
    }
    if (mi_batching_on() && b->fuse && !b->exact && !b->prof && cgraph->n_nodes != b->no_batch_nodes && mi_is_step_graph(cgraph)) return mi_batch_join(b, cgraph);
    mi_batch_leave(b);                          // anything else (encoder, prompt, beam step): this state is not decoding token by token right now
    return mi_compute_own(b, cgraph);
}

static const ggml_backend_i mi_backend_iface = {
    /* .get_name            = */ mi_backend_get_name,
    /* .free                = */ mi_backend_free,
    /* .set_tensor_async    = */ nullptr,
    /* .get_tensor_async    = */ nullptr,
    /* .set_tensor_2d_async = */ nullptr,
    /* .get_tensor_2d_async = */ nullptr,
    /* .cpy_tensor_async    = */ nullptr,
    /* .synchronize         = */ mi_backend_synchronize,
    /* .graph_plan_create   = */ nullptr,
    /* .graph_plan_free     = */ nullptr,
    /* .graph_plan_update   = */ nullptr,
    /* .graph_plan_compute  = */ nullptr,
    /* .graph_compute       = */ mi_backend_graph_compute,
    /* .event_record        = */ nullptr,
    /* .event_wait          = */ nullptr,
    /* .graph_optimize      = */ nullptr,
};

static ggml_guid_t mi_guid() {
    static ggml_guid guid = { 0x6d, 0x69, 0x33, 0x35, 0x35, 0x78, 0x2d, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x2d, 0x77, 0x31 };
    return &guid;
}

// ---------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------
static const char * mi_dev_get_name(ggml_backend_dev_t dev) { return ((mi_device_ctx *) dev->context)->name.c_str(); }
static const char * mi_dev_get_description(ggml_backend_dev_t dev) { return ((mi_device_ctx *) dev->context)->description.c_str(); }
static void mi_dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    mi_device_ctx * d = (mi_device_ctx *) dev->context;
    *free = 0; *total = 0;
    if (hipSetDevice(d->index) == hipSuccess) (void) hipMemGetInfo(free, total);
}
static enum ggml_backend_dev_type mi_dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static void mi_dev_get_props(ggml_backend_dev_t dev, ggml_backend_dev_props * props) {
    props->name = mi_dev_get_name(dev); props->description = mi_dev_get_description(dev);
    mi_dev_get_memory(dev, &props->memory_free, &props->memory_total);
    props->type = GGML_BACKEND_DEVICE_TYPE_GPU; props->device_id = nullptr;
    props->caps = { /* async */ false, /* host_buffer */ false, /* buffer_from_host_ptr */ false, /* events */ false, /* mmap */ false };
}
static ggml_backend_t mi_dev_init_backend(ggml_backend_dev_t dev, const char *) {
    mi_device_ctx * d = (mi_device_ctx *) dev->context;
    mi355x_ctx * k = mi355x_ctx_create(d->index);
    if (!k) { GGML_LOG_ERROR("ggml-mi355x: failed to create kernel context on device %d: %s\n", d->index, mi355x_last_error()); return nullptr; }
    mi_backend_ctx * b = new mi_backend_ctx();
    b->device = d->index; b->k = k; b->name = d->name;
    // Plain launches on the backend's stream.  (Rounds 1-2 carried a record / patch / replay path over hipGraphs; on ROCm 7.2 it lost to
    //  the plain launch loop on the same kernels — 1.515 vs 1.473 ms/token, profiles/r02_decode_launch_mode_sweep.txt — and was removed.)
    b->fuse = env_flag("GGML_MI355X_FUSE", true); b->prof = env_flag("GGML_MI355X_PROF", false);
    b->exact = env_flag("GGML_MI355X_EXACT", false);
    if (b->prof) mi355x_prof_enable(k, 1);
    { std::lock_guard<std::mutex> lk(g_weights_mtx); std::unique_lock<std::shared_mutex> wl(g_backends_rw); g_backends.push_back(b); }
    return new ggml_backend{ mi_guid(), mi_backend_iface, dev, b };
}
static ggml_backend_buffer_type_t mi_dev_get_buffer_type(ggml_backend_dev_t dev) { return &((mi_device_ctx *) dev->context)->buft; }
static bool mi_dev_supports_op(ggml_backend_dev_t, const ggml_tensor * op) {
    trace_scope tr(0);
    const bool ok = mi_supports_op_impl(op);
    if (!ok) {
        MI_LOG("unsupported op %s (%s) type=%s", ggml_op_name(op->op), op->name, ggml_type_name(op->type));
        // GGML_MI355X_STRICT=1 (set by bench.py and the GPU tests): a node that would silently fall back to the CPU
        // backend is a hard error, so a measured or parity-checked run is guaranteed to have executed on the HIP path
        static const bool strict = env_flag("GGML_MI355X_STRICT", false);
        if (strict) GGML_ABORT("ggml-mi355x: STRICT mode: op %s (%s, type %s) is not supported by the MI355X backend", ggml_op_name(op->op), op->name, ggml_type_name(op->type));
    }
    return ok;
}
static bool mi_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    trace_scope tr(1);
    return buft->iface.get_name == mi_buft_get_name && buft->context == dev->context;
}

static const ggml_backend_device_i mi_dev_iface = {
    /* .get_name             = */ mi_dev_get_name,
    /* .get_description      = */ mi_dev_get_description,
    /* .get_memory           = */ mi_dev_get_memory,
    /* .get_type             = */ mi_dev_get_type,
    /* .get_props            = */ mi_dev_get_props,
    /* .init_backend         = */ mi_dev_init_backend,
    /* .get_buffer_type      = */ mi_dev_get_buffer_type,
    /* .get_host_buffer_type = */ nullptr,
    /* .buffer_from_host_ptr = */ nullptr,
    /* .supports_op          = */ mi_dev_supports_op,
    /* .supports_buft        = */ mi_dev_supports_buft,
    /* .offload_op           = */ nullptr,
    /* .event_new            = */ nullptr,
    /* .event_free           = */ nullptr,
    /* .event_synchronize    = */ nullptr,
};

// ---------------------------------------------------------------------------------------------------
// registry
// ---------------------------------------------------------------------------------------------------
static void mi_init_devices() {
    static std::once_flag once;
    std::call_once(once, [] {
        int n = mi355x_device_count();
        if (n > MI_MAX_DEVICES) n = MI_MAX_DEVICES;
        g_n_devices = n;
        for (int i = 0; i < n; i++) {
            hipDeviceProp_t p;
            mi_device_ctx & d = g_device_ctx[i];
            d.index = i; d.name = "MI355X" + std::to_string(i);
            d.description = hipGetDeviceProperties(&p, i) == hipSuccess ? std::string(p.name) + " (" + p.gcnArchName + ")" : "AMD Instinct (gfx950)";
            g_devices[i] = { mi_dev_iface, &g_reg, &d };
            d.buft = { mi_buft_iface, &g_devices[i], &d };
        }
    });
}

static const char * mi_reg_get_name(ggml_backend_reg_t) { return "MI355X"; }
static size_t mi_reg_get_device_count(ggml_backend_reg_t) { mi_init_devices(); return (size_t) g_n_devices; }
static ggml_backend_dev_t mi_reg_get_device(ggml_backend_reg_t, size_t index) {
    mi_init_devices();
    GGML_ASSERT((int) index < g_n_devices);
    return &g_devices[index];
}

static ggml_mi355x_feature g_features[] = {
    { "ARCH", "gfx950" }, { "MFMA_F16", "1" }, { "DOT4_I8", "1" }, { "PLANAR_QUANT", "1" }, { nullptr, nullptr },
};

static mi_backend_ctx * as_ctx(void * backend) {
    ggml_backend_t b = (ggml_backend_t) backend;
    if (!b || b->iface.get_name != mi_backend_get_name) return nullptr;
    return (mi_backend_ctx *) b->context;
}

extern "C" {

ggml_mi355x_feature * ggml_backend_mi355x_get_features(void *) { return g_features; }

// whisper.cpp hands its n_threads to every backend of the scheduler before each graph (src/whisper.cpp:191-205, looked up by
// name).  The HIP path has no threads; the value is kept because the reference CPU flash attention splits the key range of a
// single-query step over its threads (ggml-cpu/ops.cpp:9117-9150): the reference-exact mode reproduces that chunking.
void ggml_backend_mi355x_set_n_threads(void * backend, int n_threads) {
    mi_backend_ctx * b = as_ctx(backend); if (b && n_threads > 0) b->n_threads = n_threads;
}

void ggml_backend_mi355x_prof_enable(void * backend, int on) {
    mi_backend_ctx * b = as_ctx(backend); if (!b) return;
    b->prof = on != 0; mi355x_prof_enable(b->k, on);
}
void ggml_backend_mi355x_prof_reset(void * backend) { mi_backend_ctx * b = as_ctx(backend); if (b) mi355x_prof_reset(b->k); }
int ggml_backend_mi355x_prof_report(void * backend, ggml_mi355x_prof_row * rows, int cap) {
    mi_backend_ctx * b = as_ctx(backend); if (!b) return 0;
    static_assert(sizeof(ggml_mi355x_prof_row) == sizeof(mi355x_prof_row), "row layout");
    return mi355x_prof_report(b->k, (mi355x_prof_row *) rows, cap);
}

// process-wide variants (whisper.h does not expose its ggml_backend_t handles)
void ggml_backend_mi355x_prof_enable_all(int on) {
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    for (auto * b : g_backends) { (void) hipSetDevice(b->device); b->prof = on != 0; mi355x_prof_enable(b->k, on); }
}
void ggml_backend_mi355x_prof_reset_all(void) {
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    for (auto * b : g_backends) { (void) hipSetDevice(b->device); mi355x_prof_reset(b->k); }
}
int ggml_backend_mi355x_prof_report_all(ggml_mi355x_prof_row * rows, int cap) {
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    int n = 0;
    for (auto * b : g_backends) {
        (void) hipSetDevice(b->device);
        mi355x_prof_row tmp[64];
        const int m = mi355x_prof_report(b->k, tmp, 64);
        for (int i = 0; i < m; i++) {
            int j = 0;
            for (; j < n; j++) if (!strcmp(rows[j].name, tmp[i].name)) break;
            if (j == n) { if (n >= cap) continue; rows[n].name = tmp[i].name; rows[n].calls = 0; rows[n].total_ms = rows[n].algo_bytes = rows[n].algo_flops = 0; n++; }
            rows[j].calls += tmp[i].calls; rows[j].total_ms += tmp[i].total_ms; rows[j].algo_bytes += tmp[i].algo_bytes; rows[j].algo_flops += tmp[i].algo_flops;
        }
    }
    return n;
}
// out[0] = graph_compute calls over all backends so far (out[1..3]: always 0 — the counters of the removed hipGraph replay path, kept for ABI)
void ggml_backend_mi355x_stats(uint64_t * out) {
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    for (int i = 0; i < 4; i++) out[i] = g_total_stats[i];
    for (auto * b : g_backends) out[0] += b->n_graph_compute;
}

// out[0..3] = host milliseconds spent inside graph_compute: planning (graph walk + launch recording), hipGraph node
// patching, hipGraphLaunch, eager launches — over all backends so far; out[4..7] = milliseconds inside set_tensor,
// get_tensor, cpy_tensor, synchronize; out[8..11] = their call counts; out[12] = GPU-side span (first launch .. last kernel
// done) summed over all completed graph_computes, from hipEvent pairs on the compute stream
void ggml_backend_mi355x_host_times(double * out) {
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    for (int i = 0; i < 4; i++) out[i] = g_total_host_ms[i];
    for (auto * b : g_backends) out[3] += b->t_eager_ms;
    for (int i = 0; i < 4; i++) { out[4 + i] = g_io_ns[i].load() * 1e-6; out[8 + i] = (double) g_io_calls[i].load(); }
    out[12] = g_total_gpu_span_ms;
    for (auto * b : g_backends) out[12] += b->t_gpu_span_ms;       // completed (drained) graph_computes only
}

// Device-side greedy sampling for hosts that drive whisper_decode themselves (include/mi355x_host.h): the most probable token of row `row`
// (-1: the last row) of the logits that the decoder step most recently issued BY THE CALLING THREAD left in HBM, and its margin over the
// runner-up — 16 bytes cross PCIe instead of the n_vocab floats the reference's sampler scans on the host (src/whisper.cpp:6486-6543).
// Returns the token id, or -1 when the calling thread has not run a decoder step on this plugin.
int ggml_backend_mi355x_argmax_last(int row, float * top1, float * margin) {
    mi_backend_ctx * b = t_last_backend;
    if (!b || !b->logits_dev) return -1;
    if (row < 0) row = b->logits_rows - 1;
    if (row >= b->logits_rows || hipSetDevice(b->device) != hipSuccess) return -1;
    char * md = mi_mirror_dev(b);
    if (!md) return -1;
    hipStream_t cs = (hipStream_t) mi355x_ctx_stream(b->k);
    if (b->batch_wait_stream) { (void) hipStreamWaitEvent(cs, b->batch_wait_stream, 0); b->batch_wait_stream = nullptr; }      // the step may have run as a column of a batch
    void * out_dev = md + MI_MIRROR_CAP - 64;                  // the mirror's last 64 bytes are reserved for this
    if (mi355x_argmax_top2(b->k, b->logits_dev + (size_t) row * (size_t) b->logits_n, b->logits_n, out_dev) != 0) return -1;
    b->own_dirty = true;
    if (mi355x_ctx_synchronize(b->k) != 0) return -1;
    const int32_t * r = (const int32_t *) (b->mirror_host + MI_MIRROR_CAP - 64);
    float v1, v2; memcpy(&v1, r + 1, 4); memcpy(&v2, r + 2, 4);
    if (top1) *top1 = v1;
    if (margin) *margin = v1 - v2;
    return r[0];
}

// cross-state batching (mi_batch_group): on = 1 / 0 at run time (the environment's GGML_MI355X_BATCH is only the initial value)
void ggml_backend_mi355x_set_batching(int on) { g_batching.store(on > 0 ? on : 0); }
int  ggml_backend_mi355x_get_batching(void) { (void) mi_batching_on(); return g_batching.load(); }
// out[0..4] of `device`: merged launch chains, columns they carried, steps a state ran alone, groups that fell back to one launch chain
// per state (graphs did not fit), windows that closed on an absent state
void ggml_backend_mi355x_batch_stats(int device, uint64_t * out5) {
    for (int i = 0; i < 5; i++) out5[i] = 0;
    if (device < 0 || device >= MI_MAX_DEVICES) return;
    mi_batch_group & grp = g_batch[device];
    std::lock_guard<std::mutex> lk(grp.m);
    out5[0] = grp.n_batches; out5[1] = grp.n_columns; out5[2] = grp.n_solo; out5[3] = grp.n_fallback.load(); out5[4] = grp.n_timeouts;
}

// TEST hook (no device needed: nothing is launched).  S >= 2: would S copies of `cgraph` run as the columns of one launch chain?
// out[0] = mi_walk_batch's verdict (0 yes).  S == 1: which nodes would the T >= 3 plane pipeline take?  out[1..3] = LayerNorm stages,
// attention stages, plain mat-vec stages taken; out[4] = compute nodes left to the other paths; out[5] = stages whose epilogue writes planes.
int ggml_backend_mi355x_debug_walk(void * cgraph, int S, int64_t * out6) {
    ggml_cgraph * g = (ggml_cgraph *) cgraph;
    for (int i = 0; i < 6; i++) out6[i] = 0;
    if (S >= 2) {
        if (S > MI355X_MAX_COLS) return -1;
        mi_colset cs; cs.S = S; cs.T = S;
        for (int c = 0; c < S; c++) cs.g[c] = g;
        out6[0] = mi_is_step_graph(g) && mi_graphs_congruent(g, g) ? mi_walk_batch(nullptr, cs) : -2;
        return 0;
    }
    mi_qstate qs;
    for (int i = 0; i < g->n_nodes; i++) {
        const ggml_tensor * n = g->nodes[i];
        if (op_is_empty(n) || ggml_is_empty(n) || !(n->flags & GGML_TENSOR_FLAG_COMPUTE)) continue;
        int end = i, rc = 0; bool took = false;
        if (n->op == GGML_OP_MUL_MAT || n->op == GGML_OP_NORM || n->op == GGML_OP_FLASH_ATTN_EXT) {
            const int64_t Tn = n->op == GGML_OP_FLASH_ATTN_EXT ? n->src[0]->ne[1] : (n->op == GGML_OP_MUL_MAT ? n->src[1]->ne[1] : ggml_nrows(n->src[0]));
            if (Tn >= 1 && Tn <= MI355X_IMG_COLS) {
                mi_colset cs; cs.S = 1; cs.T = (int) Tn; cs.g[0] = g;
                if (n->op == GGML_OP_NORM) { ln_chain c; parse_ln_chain(g, i, true, c); took = q_ln_gemv(nullptr, cs, qs, i, c, end, rc); if (took) out6[1]++; }
                else if (n->op == GGML_OP_FLASH_ATTN_EXT) { took = q_attn_proj(nullptr, cs, qs, i, end, rc); if (took) out6[2]++; }
                else { took = q_mm(nullptr, cs, qs, i, end, rc); if (took) out6[3]++; }
            }
        }
        if (took) i = end; else out6[4]++;
    }
    return 0;
}

// GGML_MI355X_TRACE=1: out[2*i] = nanoseconds, out[2*i + 1] = calls of trace slot i (8 slots, see g_trace_ns); returns 0 when tracing is off
int ggml_backend_mi355x_trace(uint64_t * out16) {
    for (int i = 0; i < 8; i++) { out16[2*i] = g_trace_ns[i].load(); out16[2*i + 1] = g_trace_calls[i].load(); }
    return g_trace ? 1 : 0;
}

int ggml_backend_mi355x_weight_buffers(int device, void ** bases, size_t * sizes, int cap) {
    mi_shadows_drop(device, nullptr);          // the caller is about to overwrite the weights behind our back
    if (hipSetDevice(device) == hipSuccess) { mi_io_drain(device); (void) hipDeviceSynchronize(); }     // small weights uploaded through the pinned ring have landed
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    int n = 0;
    for (auto & r : g_buffers) {
        if (r.device != device || r.buf->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS) continue;
        if (n < cap) { bases[n] = r.base; sizes[n] = r.size; }
        n++;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU weight distribution: replicas are independent streams, the only exchange is the ONE-TIME copy
// of rank 0's WEIGHTS buffers into the identically laid out buffers of the other replicas (every context allocates the same
// tensors in the same order, src/whisper.cpp:1685-1859, so buffer i has the same size everywhere — checked).  Two transports:
//   * one process, several devices   : hipMemcpyPeerAsync (xGMI peer copy)                  ggml_backend_mi355x_broadcast_weights_peer
//   * one process per device (torchrun): RCCL ncclBroadcast on a communicator built here from a 128-byte unique id that the host
//     harness hands to every rank (librccl.so is dlopen()ed: the plugin does not link it)   ggml_backend_mi355x_broadcast_weights_rccl
// Either way every buffer is then check-summed on the device (mi355x_checksum) and compared with the source's: a replica that does
// not hold rank 0's bytes is an error, never a silently different model.
// ---------------------------------------------------------------------------------------------------
static std::vector<mi_weight_rec> mi_weight_list(int device) {
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    std::vector<mi_weight_rec> v;
    for (auto & r : g_buffers) if (r.device == device && r.buf->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS) v.push_back(r);
    return v;
}

static int mi_checksums(int device, const std::vector<mi_weight_rec> & bufs, std::vector<uint64_t> & sums) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    void * d = nullptr;
    if (hipMalloc(&d, 16) != hipSuccess) return -1;
    sums.assign(bufs.size() * 2, 0);
    int rc = 0;
    for (size_t i = 0; i < bufs.size() && rc == 0; i++) {
        if (mi355x_checksum(nullptr, bufs[i].base, bufs[i].size, d) != 0 || hipMemcpy(&sums[2*i], d, 16, hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    }
    (void) hipFree(d);
    return rc;
}

#include <dlfcn.h>
// RCCL is dlopen()ed; only a handful of its types are needed here.  With the RCCL headers installed they come from there, on a ROCm
// install without them the same (ABI-stable, nccl.h) declarations are made locally so that the plugin still builds.
#if defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm * ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
#endif

struct mi_rccl_api {
    void * h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char * (*GetErrorString)(ncclResult_t) = nullptr;
};
static mi_rccl_api * mi_rccl() {
    static mi_rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char * n : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.h) break; }
        if (!api.h) return;
        api.GetUniqueId    = (decltype(api.GetUniqueId))    dlsym(api.h, "ncclGetUniqueId");
        api.CommInitRank   = (decltype(api.CommInitRank))   dlsym(api.h, "ncclCommInitRank");
        api.CommDestroy    = (decltype(api.CommDestroy))    dlsym(api.h, "ncclCommDestroy");
        api.Broadcast      = (decltype(api.Broadcast))      dlsym(api.h, "ncclBroadcast");
        api.AllReduce      = (decltype(api.AllReduce))      dlsym(api.h, "ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString)) dlsym(api.h, "ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Broadcast || !api.AllReduce) { dlclose(api.h); api.h = nullptr; }
    });
    return api.h ? &api : nullptr;
}

extern "C" {

// uploads of weight tensors are skipped while the flag is set (replicas whose weights will arrive by broadcast: the skipping
// model loader of the host harness never reads tensor payloads, this covers callers that do)
void ggml_backend_mi355x_defer_weights(int on) { t_defer_weights = on ? 1 : 0; }
uint64_t ggml_backend_mi355x_deferred_bytes(void) { return g_deferred_bytes.load(); }

// out[0..2n): {sum, weighted sum} of every WEIGHTS buffer of `device` in allocation order; returns n (or -1)
int ggml_backend_mi355x_weights_checksum(int device, uint64_t * out, int cap) {
    mi_shadows_drop(device, nullptr);
    if (hipSetDevice(device) == hipSuccess) { mi_io_drain(device); (void) hipDeviceSynchronize(); }
    const std::vector<mi_weight_rec> bufs = mi_weight_list(device);
    std::vector<uint64_t> sums;
    if (mi_checksums(device, bufs, sums) != 0) return -1;
    for (size_t i = 0; i < bufs.size() && (int) i < cap; i++) { out[2*i] = sums[2*i]; out[2*i + 1] = sums[2*i + 1]; }
    return (int) bufs.size();
}

// in-process: copy every WEIGHTS buffer of src_device into the same-index buffer of dst_device, then verify.
// stats[0..3] = bytes, seconds, buffers, verified (1/0).  Returns 0, or a negative code (-2 layout mismatch, -3 copy failed, -4 checksum mismatch).
static int mi_copy_verify(int src_device, int dst_device, const std::vector<mi_weight_rec> & S, const std::vector<mi_weight_rec> & D, double * stats) {
    if (S.empty() || S.size() != D.size()) return -2;
    for (size_t i = 0; i < S.size(); i++) if (S[i].size != D[i].size) return -2;
    for (int dev : { src_device, dst_device }) { if (hipSetDevice(dev) != hipSuccess) return -3; mi_io_drain(dev); (void) hipDeviceSynchronize(); }
    if (src_device != dst_device) {
        int can = 0;
        (void) hipDeviceCanAccessPeer(&can, dst_device, src_device);
        if (can) { (void) hipSetDevice(dst_device); hipError_t e = hipDeviceEnablePeerAccess(src_device, 0); if (e != hipSuccess) (void) hipGetLastError(); }
    }
    const double t0 = now_ms();
    double bytes = 0;
    (void) hipSetDevice(dst_device);
    for (size_t i = 0; i < S.size(); i++) {
        const hipError_t e = src_device != dst_device ? hipMemcpyPeerAsync(D[i].base, dst_device, S[i].base, src_device, S[i].size, nullptr)
                                                      : hipMemcpyAsync(D[i].base, S[i].base, S[i].size, hipMemcpyDeviceToDevice, nullptr);
        if (e != hipSuccess) return -3;
        bytes += (double) S[i].size;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    const double secs = (now_ms() - t0) * 1e-3;
    std::vector<uint64_t> cs, cd;
    if (mi_checksums(src_device, S, cs) != 0 || mi_checksums(dst_device, D, cd) != 0) return -4;
    const bool ok = cs == cd;
    if (stats) { stats[0] += bytes; stats[1] += secs; stats[2] = (double) S.size(); stats[3] = ok ? 1 : 0; }
    return ok ? 0 : -4;
}

int ggml_backend_mi355x_broadcast_weights_peer(int src_device, int dst_device, double * stats) {
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (src_device == dst_device) return -2;
    mi_shadows_drop(dst_device, nullptr);
    return mi_copy_verify(src_device, dst_device, mi_weight_list(src_device), mi_weight_list(dst_device), stats);
}

// n_replicas contexts created one after the other on ONE device (a one-GPU machine standing in for n GPUs, so that the
// payload-skipping load -> copy -> verify -> run path can be executed on real hardware): the device's WEIGHTS buffers are
// n_replicas groups of k in allocation order; group 0 is copied into every other group and verified like a peer broadcast.
int ggml_backend_mi355x_clone_weights(int device, int n_replicas, double * stats) {
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    mi_shadows_drop(device, nullptr);
    const std::vector<mi_weight_rec> all = mi_weight_list(device);
    if (n_replicas < 2 || all.empty() || all.size() % (size_t) n_replicas) return -2;
    const size_t k = all.size() / (size_t) n_replicas;
    const std::vector<mi_weight_rec> S(all.begin(), all.begin() + k);
    for (int g = 1; g < n_replicas; g++) {
        const std::vector<mi_weight_rec> D(all.begin() + g * k, all.begin() + (g + 1) * k);
        const int rc = mi_copy_verify(device, device, S, D, stats);
        if (rc != 0) return rc;
    }
    return 0;
}

int ggml_backend_mi355x_rccl_unique_id(void * out128) {
    mi_rccl_api * r = mi_rccl();
    if (!r) return -1;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return -1;
    memcpy(out128, &id, sizeof(id));
    return 0;
}

// one process per device: ncclBroadcast of every WEIGHTS buffer of `device` from rank 0 over a communicator created here from the
// shared unique id, then a checksum of every buffer compared across ALL ranks (allreduce min / max must agree).
// stats[0..3] as above.  Returns 0 or a negative code (-1 RCCL unavailable / failed, -2 layout mismatch between ranks, -4 checksum mismatch).
int ggml_backend_mi355x_broadcast_weights_rccl(int device, int rank, int world, const void * unique_id128, double * stats) {
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    mi_rccl_api * r = mi_rccl();
    if (!r || world < 1 || rank < 0 || rank >= world) return -1;
    mi_shadows_drop(device, nullptr);
    if (hipSetDevice(device) != hipSuccess) return -1;
    mi_io_drain(device); (void) hipDeviceSynchronize();
    const std::vector<mi_weight_rec> B = mi_weight_list(device);
    ncclUniqueId id; memcpy(&id, unique_id128, sizeof(id));
    // Local resources FIRST: a rank that cannot get them still joins the communicator and the agreement round below, so that no other
    // rank is left waiting inside a collective for it.  (A rank whose ncclCommInitRank itself fails cannot be helped from here: the
    // others block in their own init until RCCL's bootstrap times out — the host harness's all_ranks_ok round then reports it.)
    hipStream_t st = nullptr;
    uint64_t * d64 = nullptr;          // device scratch: layout words, then checksums
    const size_t nw = 1 + B.size();
    const size_t cap_words = 3 * std::max(nw, 2 * B.size() + 2) + 8;
    bool local_ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    if (hipMalloc((void **) &d64, cap_words * 8) != hipSuccess) { d64 = nullptr; local_ok = false; (void) hipGetLastError(); }
    ncclComm_t comm = nullptr;
    if (r->CommInitRank(&comm, world, id, rank) != ncclSuccess) {
        if (d64) (void) hipFree(d64);
        if (st) (void) hipStreamDestroy(st);
        return -1;
    }
    int rc = 0;
    std::vector<uint64_t> lay(nw), lo(nw), hi(nw);
    lay[0] = B.size(); for (size_t i = 0; i < B.size(); i++) lay[1 + i] = B[i].size;
    auto all_equal = [&](std::vector<uint64_t> & v) -> bool {      // every rank holds the same words?  (min == max over ranks)
        const size_t n = v.size();
        if (hipMemcpy(d64, v.data(), n*8, hipMemcpyHostToDevice) != hipSuccess) return false;
        if (r->AllReduce(d64, d64 + n, n, ncclUint64, ncclMin, comm, st) != ncclSuccess) return false;
        if (r->AllReduce(d64, d64 + 2*n, n, ncclUint64, ncclMax, comm, st) != ncclSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        std::vector<uint64_t> a(n), b(n);
        if (hipMemcpy(a.data(), d64 + n, n*8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(b.data(), d64 + 2*n, n*8, hipMemcpyDeviceToHost) != hipSuccess) return false;
        return a == b;
    };
    if (!local_ok) {
        // this rank cannot run the data collectives: it cannot take part in the agreement either (no device word), so it leaves;
        // the communicator is destroyed, which makes the peers' next collective fail instead of hang
        if (d64) (void) hipFree(d64);
        if (st) (void) hipStreamDestroy(st);
        (void) r->CommDestroy(comm);
        return -1;
    }
    if (rc == 0 && !all_equal(lay)) rc = -2;                        // same number of buffers, same sizes, on every rank
    double bytes = 0, secs = 0;
    if (rc == 0) {
        const double t0 = now_ms();
        for (size_t i = 0; i < B.size() && rc == 0; i++) {
            if (r->Broadcast(B[i].base, B[i].base, B[i].size, ncclUint8, 0, comm, st) != ncclSuccess) rc = -1;
            bytes += (double) B[i].size;
        }
        if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = -1;
        secs = (now_ms() - t0) * 1e-3;
    }
    bool verified = false;
    if (rc != -2) {
        // every rank that agreed on the layout takes part in this round, also one whose broadcast or checksum failed locally: it
        // contributes words that cannot match (its rank in the high bits), so ALL ranks see "not verified" and nobody waits forever
        std::vector<uint64_t> cs;
        const bool local = rc == 0 && mi_checksums(device, B, cs) == 0;
        if (!local) cs.assign(2 * B.size(), 0xBAD0000000000000ull | (uint64_t) rank);
        cs.push_back(0xC0FFEEull); cs.push_back((uint64_t) B.size());
        verified = all_equal(cs) && local;
        if (!verified && rc == 0) rc = -4;
    }
    if (d64) (void) hipFree(d64);
    if (st) (void) hipStreamDestroy(st);
    (void) r->CommDestroy(comm);
    if (stats) { stats[0] = bytes; stats[1] = secs; stats[2] = (double) B.size(); stats[3] = verified ? 1 : 0; }
    return rc;
}

} // extern "C"

static void * mi_reg_get_proc_address(ggml_backend_reg_t, const char * name) {
    trace_scope tr(5);
    if (!strcmp(name, "ggml_backend_get_features"))           return (void *) ggml_backend_mi355x_get_features;
    if (!strcmp(name, "ggml_backend_set_n_threads"))          return (void *) ggml_backend_mi355x_set_n_threads;
    if (!strcmp(name, "ggml_backend_mi355x_prof_enable"))     return (void *) ggml_backend_mi355x_prof_enable;
    if (!strcmp(name, "ggml_backend_mi355x_prof_reset"))      return (void *) ggml_backend_mi355x_prof_reset;
    if (!strcmp(name, "ggml_backend_mi355x_prof_report"))     return (void *) ggml_backend_mi355x_prof_report;
    if (!strcmp(name, "ggml_backend_mi355x_weight_buffers"))  return (void *) ggml_backend_mi355x_weight_buffers;
    if (!strcmp(name, "ggml_backend_mi355x_weights_checksum"))        return (void *) ggml_backend_mi355x_weights_checksum;
    if (!strcmp(name, "ggml_backend_mi355x_broadcast_weights_peer"))  return (void *) ggml_backend_mi355x_broadcast_weights_peer;
    if (!strcmp(name, "ggml_backend_mi355x_clone_weights"))           return (void *) ggml_backend_mi355x_clone_weights;
    if (!strcmp(name, "ggml_backend_mi355x_broadcast_weights_rccl"))  return (void *) ggml_backend_mi355x_broadcast_weights_rccl;
    if (!strcmp(name, "ggml_backend_mi355x_rccl_unique_id"))          return (void *) ggml_backend_mi355x_rccl_unique_id;
    if (!strcmp(name, "ggml_backend_mi355x_defer_weights"))           return (void *) ggml_backend_mi355x_defer_weights;
    if (!strcmp(name, "ggml_backend_mi355x_deferred_bytes"))          return (void *) ggml_backend_mi355x_deferred_bytes;
    if (!strcmp(name, "ggml_backend_mi355x_prof_enable_all")) return (void *) ggml_backend_mi355x_prof_enable_all;
    if (!strcmp(name, "ggml_backend_mi355x_prof_reset_all"))  return (void *) ggml_backend_mi355x_prof_reset_all;
    if (!strcmp(name, "ggml_backend_mi355x_prof_report_all")) return (void *) ggml_backend_mi355x_prof_report_all;
    if (!strcmp(name, "ggml_backend_mi355x_stats"))           return (void *) ggml_backend_mi355x_stats;
    if (!strcmp(name, "ggml_backend_mi355x_host_times"))      return (void *) ggml_backend_mi355x_host_times;
    if (!strcmp(name, "ggml_backend_mi355x_trace"))           return (void *) ggml_backend_mi355x_trace;
    if (!strcmp(name, "ggml_backend_mi355x_set_batching"))    return (void *) ggml_backend_mi355x_set_batching;
    if (!strcmp(name, "ggml_backend_mi355x_get_batching"))    return (void *) ggml_backend_mi355x_get_batching;
    if (!strcmp(name, "ggml_backend_mi355x_argmax_last"))     return (void *) ggml_backend_mi355x_argmax_last;
    if (!strcmp(name, "ggml_backend_mi355x_debug_walk"))      return (void *) ggml_backend_mi355x_debug_walk;
    if (!strcmp(name, "ggml_backend_mi355x_batch_stats"))     return (void *) ggml_backend_mi355x_batch_stats;
    return nullptr;
}

static const ggml_backend_reg_i mi_reg_iface = {
    /* .get_name         = */ mi_reg_get_name,
    /* .get_device_count = */ mi_reg_get_device_count,
    /* .get_device       = */ mi_reg_get_device,
    /* .get_proc_address = */ mi_reg_get_proc_address,
};

void * ggml_backend_mi355x_reg(void) {
    static std::once_flag once;
    std::call_once(once, [] { g_reg = { GGML_BACKEND_API_VERSION, mi_reg_iface, nullptr }; });
    return &g_reg;
}

void * ggml_backend_init(void) { return ggml_backend_mi355x_reg(); }

int ggml_backend_score(void) { return mi355x_device_count() > 0 ? 100 : 0; }

} // extern "C"
