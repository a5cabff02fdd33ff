// mi_buffers.cpp — part of the MI355X ggml backend plugin; see mi_backend.h for the map of the translation units.
#include "mi_backend.h"



std::mutex                 g_weights_mtx;
std::vector<mi_weight_rec> g_buffers;       // every live device buffer, in allocation order

ggml_backend_reg           g_reg;
ggml_backend_device        g_devices[MI_MAX_DEVICES];
mi_device_ctx              g_device_ctx[MI_MAX_DEVICES];
int                        g_n_devices = -1;

bool is_quant_type(ggml_type t) { return mi355x_type_is_quantized((int) t) != 0; }
// ggml_backend_mi355x_defer_weights: weight uploads issued BY THE CALLING THREAD are skipped (they arrive by broadcast).  Scoped to
// the thread that creates the replica's context — whisper_init_* runs its set_tensor loop on the caller's thread (src/whisper.cpp:1934-
// 1938) — so another context loading concurrently on another thread, or on another device, is never affected.
thread_local int       t_defer_weights = 0;
std::atomic<uint64_t>  g_deferred_bytes{0};    // bytes skipped so far (visible through GGML_MI355X_DEBUG and ggml_backend_mi355x_deferred_bytes)

std::mutex                                   g_shadow_mtx;
std::unordered_map<const void *, mi_shadow>  g_shadows;
std::atomic<size_t>                          g_shadow_count{0};
size_t                                       g_shadow_bytes = 0;

// forget (and free) the copies that belong to one buffer (buf_base) or one device (buf_base == nullptr)
void mi_shadows_drop(int device, const void * buf_base) {
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
std::atomic<uint64_t> g_io_ns[4] = {};      // set_tensor, get_tensor, cpy_tensor, synchronize
std::atomic<uint64_t> g_io_calls[4] = {};

// GGML_MI355X_TRACE=1: where the host-side time of a step goes INSIDE the plugin (tests/native/step_trace.cpp reads it through
// ggml_backend_mi355x_trace and sets it beside the reference-side timeline it measures by interposing the scheduler's entry points).
// Slots (ns, calls): 0 supports_op, 1 supports_buft, 2 graph_compute entry -> first kernel launched, 3 graph_compute entry -> return,
// 4 graph_compute entry -> synchronize return (one step's whole device phase as the host sees it), 5 get_proc_address
const bool g_trace = env_flag("GGML_MI355X_TRACE", false);
std::atomic<uint64_t> g_trace_ns[8] = {};
std::atomic<uint64_t> g_trace_calls[8] = {};

mi_io_ctx g_io[MI_MAX_DEVICES];

mi_io_ctx * mi_io(int device) {          // caller holds no lock; device already current
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
void mi_io_flush_locked(mi_io_ctx & io, hipStream_t stream) {
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
void mi_io_drain(int device) {
    mi_io_ctx & io = g_io[device];
    if (!io.ok || io.drained.load() == io.seq.load()) return;
    std::lock_guard<std::mutex> lk(io.mtx);
    const uint64_t s = io.seq.load();
    mi_io_flush_locked(io, io.stream);
    (void) hipStreamSynchronize(io.stream);
    if (io.flushed_since_drain) { (void) hipEventSynchronize(io.ev_flush); io.flushed_since_drain = false; }
    io.drained.store(s); io.off = 0;
}
bool mi_io_upload(int device, void * dst, const void * src, size_t size) {
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
bool mi_io_download(int device, void * dst, const void * src, size_t size) {
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


// ---------------------------------------------------------------------------------------------------
// buffer
// ---------------------------------------------------------------------------------------------------
void mi_buffer_free(ggml_backend_buffer_t buffer) {
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

void * mi_buffer_get_base(ggml_backend_buffer_t buffer) { return ((mi_buffer_ctx *) buffer->context)->base; }

// quantized tensors are stored planar (include/mi355x_kernels.h); whole-tensor transfers re-layout on the host,
// partial ones go through read-modify-write of the whole tensor (never happens in whisper.cpp: W:1934-1938)
// The planar layout is defined per WHOLE tensor (the planes of NB blocks follow each other): a row / sub-view of a quantized
// tensor has no contiguous image in it, and a non-contiguous one cannot be re-laid out block by block.  whisper.cpp only ever
// transfers whole weight tensors (W:1934-1938); anything else is a programmer error and must not silently corrupt weights.
bool whole_quant_tensor(const ggml_tensor * t) {
    return ggml_is_contiguous(t) && (!t->view_src || (t->view_offs == 0 && ggml_nbytes(t) == ggml_nbytes(t->view_src)));
}

void mi_buffer_set_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
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

void mi_buffer_get_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * tensor, void * data, size_t offset, size_t size) {
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

void mi_buffer_memset_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
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

void mi_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) {
    mi_buffer_ctx * ctx = (mi_buffer_ctx *) buffer->context;
    mi_shadows_drop(ctx->device, ctx->base);
    mi_mirror_invalidate(ctx->device, ctx->base, ctx->size);
    (void) hipSetDevice(ctx->device);
    mi_io_drain(ctx->device);
    (void) hipMemset(ctx->base, value, ctx->size);
    (void) hipDeviceSynchronize();
}

bool mi_buffer_is_ours(ggml_backend_buffer_t buffer) { return buffer && buffer->iface.get_base == mi_buffer_get_base; }

bool mi_buffer_cpy_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * src, ggml_tensor * dst) {
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

const ggml_backend_buffer_i mi_buffer_iface = {
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
const char * mi_buft_get_name(ggml_backend_buffer_type_t buft) { return ((mi_device_ctx *) buft->context)->name.c_str(); }

ggml_backend_buffer_t mi_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
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
size_t mi_buft_get_alignment(ggml_backend_buffer_type_t) { return MI_ALIGNMENT; }
size_t mi_buft_get_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * tensor) { return ggml_nbytes(tensor); }
bool   mi_buft_is_host(ggml_backend_buffer_type_t) { return false; }

const ggml_backend_buffer_type_i mi_buft_iface = {
    /* .get_name       = */ mi_buft_get_name,
    /* .alloc_buffer   = */ mi_buft_alloc_buffer,
    /* .get_alignment  = */ mi_buft_get_alignment,
    /* .get_max_size   = */ nullptr,
    /* .get_alloc_size = */ mi_buft_get_alloc_size,
    /* .is_host        = */ mi_buft_is_host,
};


double g_total_gpu_span_ms = 0;

void mi_span_drain(mi_backend_ctx * b) {          // all pending pairs must have completed (caller synchronized the stream)
    for (int i = 0; i < b->span_pending; i++) {
        const int idx = (b->span_next - 1 - i + 2 * (int) b->span_ev.size()) % (int) b->span_ev.size();
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->span_ev[idx].first, b->span_ev[idx].second) == hipSuccess) b->t_gpu_span_ms += ms;
    }
    b->span_pending = 0;
}

void mi_io_snapshot(int device, const char * const * base, const size_t * size, int n, mi_io_saved & out) {
    out.recs.clear(); out.data.clear();
    mi_io_ctx & io = g_io[device];
    if (!io.ok) return;
    std::lock_guard<std::mutex> lk(io.mtx);
    for (const mi_io_rec & r : io.pending) {
        bool mine = false;
        for (int i = 0; i < n && !mine; i++) mine = (const char *) r.dst >= base[i] && (const char *) r.dst + r.size <= base[i] + size[i];
        if (!mine) continue;
        out.recs.push_back({ r.dst, (uint32_t) out.data.size(), r.size });
        out.data.insert(out.data.end(), io.pinned + r.off, io.pinned + r.off + r.size);
    }
}
void mi_io_replay(int device, const mi_io_saved & s) {
    for (const mi_io_rec & r : s.recs)
        if (!mi_io_upload(device, r.dst, s.data.data() + r.off, r.size)) GGML_ABORT("ggml-mi355x: re-staging the graph inputs of a rejected chain failed");
}

void mi_io_order_stream(int device, mi_io_marks & mk, hipStream_t cs) {
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


std::vector<mi_backend_ctx *> g_backends;           // live backends (guarded by g_weights_mtx; the mirror look-ups below only share g_backends_rw)
std::shared_mutex g_backends_rw;                     // writers (backend init / free) hold it exclusively IN ADDITION to g_weights_mtx: every stream's per-step
                                                            // uploads and logits reads scan the list, and must not serialise on one process-wide mutex
thread_local mi_backend_ctx * t_last_backend = nullptr;    // the backend whose graph_compute this host thread called last (one thread per whisper_state)

const bool g_mirror_on = env_flag("GGML_MI355X_LOGITS_MIRROR", true);
char * mi_mirror_dev(mi_backend_ctx * b) {             // device address of the backend's mirror (allocated on first use), or nullptr
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
void mi_mirror_invalidate(int device, const void * p, size_t n) {
    std::shared_lock<std::shared_mutex> lk(g_backends_rw);
    for (auto * b : g_backends)
        if (b->device == device && b->mirror_state.load() != 0 && (const char *) b->mirror_src.load() < (const char *) p + n && (const char *) p < (const char *) b->mirror_src.load() + b->mirror_bytes.load()) b->mirror_state.store(0);
}
// read [src, src + size) from a valid mirror instead of the device; false: no mirror holds it
bool mi_mirror_read(int device, const void * src, void * dst, size_t size) {
    std::shared_lock<std::shared_mutex> lk(g_backends_rw);          // (shared: several streams copy their rows at the same time)
    for (auto * b : g_backends) {
        if (b->device != device || b->mirror_state.load() != 2) continue;
        const char * s0 = (const char *) b->mirror_src.load();
        if ((const char *) src >= s0 && (const char *) src + size <= s0 + b->mirror_bytes.load()) { memcpy(dst, b->mirror_host + ((const char *) src - s0), size); return true; }
    }
    return false;
}
uint64_t g_total_stats[4] = { 0, 0, 0, 0 };         // counters of already freed backends
double   g_total_host_ms[4] = { 0, 0, 0, 0 };

