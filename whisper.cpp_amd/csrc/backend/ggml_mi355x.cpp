// ggml backend plugin for AMD Instinct MI355X (gfx950).  Host side only: registration, buffers, graph walk +
// fusion planner.  All device work goes through the C ABI of libmi355x_kernels.so
// (include/mi355x_kernels.h).  Boundary documentation: include/ggml_mi355x.h.
//
// Reference interface being implemented: ggml/src/ggml-backend-impl.h (vtables), loader
#include "mi_backend.h"

const char * mi_backend_get_name(ggml_backend_t backend) { return ((mi_backend_ctx *) backend->context)->name.c_str(); }


void mi_backend_free(ggml_backend_t backend) {
    mi_backend_ctx * b = (mi_backend_ctx *) backend->context;
    (void) hipSetDevice(b->device);
    mi_batch_leave(b);
    if (t_last_backend == b) t_last_backend = nullptr;
    mi355x_ctx_synchronize(b->k);
    if (b->batch_wait_sync) (void) hipEventSynchronize(b->batch_wait_sync);
    if (b->own_ev) (void) hipEventDestroy(b->own_ev);
    b->mirror_state.store(0);
    if (g_debug()) fprintf(stderr, "ggml-mi355x: backend %s: graph_compute=%" PRIu64 "\n", b->name.c_str(), (uint64_t) b->n_graph_compute.load());
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

void mi_backend_synchronize(ggml_backend_t backend) {
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
ggml_status mi_compute_own(mi_backend_ctx * b, ggml_cgraph * cgraph) {
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
ggml_status mi_backend_graph_compute(ggml_backend_t backend, ggml_cgraph * cgraph) {
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
    if (!mi_batching_on() && mi_is_step_graph(cgraph)) {
        // cross-state batching switched off (GGML_MI355X_BATCH=0 / ggml_backend_mi355x_set_batching(0)): beyond four decoding states their own chains collapse
        // (8 states: 2.1-2.5 chunks/s of large-v3 Q5_0 against 2.8 for ONE state and 10 merged) — say so once instead of silently being slow
        static std::atomic<bool> warned{false};
        if (!warned.load(std::memory_order_relaxed)) {
            int live = 0;
            { std::shared_lock<std::shared_mutex> rl(g_backends_rw); for (mi_backend_ctx * o : g_backends) if (o->device == b->device && o->n_graph_compute > 16) live++; }
            if (live > 4 && !warned.exchange(true))
                GGML_LOG_WARN("ggml-mi355x: %d whisper_states decode on device %d with cross-state batching OFF: beyond 4 states their launch chains serialize and throughput falls below one state's; "
                              "leave GGML_MI355X_BATCH at its default (1) for merged chains\n", live, b->device);
        }
    }
    mi_batch_leave(b);                          // anything else (encoder, prompt, beam step): this state is not decoding token by token right now
    return mi_compute_own(b, cgraph);
}

const ggml_backend_i mi_backend_iface = {
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

ggml_guid_t mi_guid() {
    static ggml_guid guid = { 0x6d, 0x69, 0x33, 0x35, 0x35, 0x78, 0x2d, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x2d, 0x77, 0x31 };
    return &guid;
}

// ---------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------
const char * mi_dev_get_name(ggml_backend_dev_t dev) { return ((mi_device_ctx *) dev->context)->name.c_str(); }
const char * mi_dev_get_description(ggml_backend_dev_t dev) { return ((mi_device_ctx *) dev->context)->description.c_str(); }
void mi_dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    mi_device_ctx * d = (mi_device_ctx *) dev->context;
    *free = 0; *total = 0;
    if (hipSetDevice(d->index) == hipSuccess) (void) hipMemGetInfo(free, total);
}
enum ggml_backend_dev_type mi_dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
void mi_dev_get_props(ggml_backend_dev_t dev, ggml_backend_dev_props * props) {
    props->name = mi_dev_get_name(dev); props->description = mi_dev_get_description(dev);
    mi_dev_get_memory(dev, &props->memory_free, &props->memory_total);
    props->type = GGML_BACKEND_DEVICE_TYPE_GPU; props->device_id = nullptr;
    props->caps = { /* async */ false, /* host_buffer */ false, /* buffer_from_host_ptr */ false, /* events */ false, /* mmap */ false };
}
ggml_backend_t mi_dev_init_backend(ggml_backend_dev_t dev, const char *) {
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
    if (mi_mmq_mode() == 3) mi355x_test_option(MI355X_OPT_DQ_GEMM, 1, 1);      // GGML_MI355X_MMQ=3: quantized weight x wide f16 activations on k_gemm_dq (process-wide kernel-library switch)
    { std::lock_guard<std::mutex> lk(g_weights_mtx); std::unique_lock<std::shared_mutex> wl(g_backends_rw); g_backends.push_back(b); }
    return new ggml_backend{ mi_guid(), mi_backend_iface, dev, b };
}
ggml_backend_buffer_type_t mi_dev_get_buffer_type(ggml_backend_dev_t dev) { return &((mi_device_ctx *) dev->context)->buft; }
bool mi_dev_supports_op(ggml_backend_dev_t, const ggml_tensor * op) {
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
bool mi_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    trace_scope tr(1);
    return buft->iface.get_name == mi_buft_get_name && buft->context == dev->context;
}

const ggml_backend_device_i mi_dev_iface = {
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
void mi_init_devices() {
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

const char * mi_reg_get_name(ggml_backend_reg_t) { return "MI355X"; }
size_t mi_reg_get_device_count(ggml_backend_reg_t) { mi_init_devices(); return (size_t) g_n_devices; }
ggml_backend_dev_t mi_reg_get_device(ggml_backend_reg_t, size_t index) {
    mi_init_devices();
    GGML_ASSERT((int) index < g_n_devices);
    return &g_devices[index];
}

ggml_mi355x_feature g_features[] = {
    { "ARCH", "gfx950" }, { "MFMA_F16", "1" }, { "DOT4_I8", "1" }, { "PLANAR_QUANT", "1" }, { nullptr, nullptr },
};

mi_backend_ctx * as_ctx(void * backend) {
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
    if (!strcmp(name, "ggml_backend_mi355x_broadcast_weights_rccl_group")) return (void *) ggml_backend_mi355x_broadcast_weights_rccl_group;
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

// One hardware queue per concurrent stream.  ROCm's default is 4 per process (incl. the upload stream): with more than 3 whisper_states decoding
// on their own chains (whisper_full_parallel, a server with several states) two HIP streams share a queue and serialize (measured: 4 streams
// 4.7 -> 5.8 chunks/s with 8 queues; HISTORY.md).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the
// process, which for a stock whisper.cpp host comes after this library has been loaded; a value the user has set is left alone.
__attribute__((constructor)) static void mi_env_defaults(void) { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

int ggml_backend_score(void) { return mi355x_device_count() > 0 ? 100 : 0; }

} // extern "C"

