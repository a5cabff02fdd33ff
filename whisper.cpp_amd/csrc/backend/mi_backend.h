// Internal header of the MI355X ggml backend plugin (libggml-mi355x.so): the structures, globals and functions its translation units share.
//   ggml_mi355x.cpp       backend / device / registry vtables (ggml-backend-impl.h) and the exported C API (include/ggml_mi355x.h)
//   mi_buffers.cpp        device buffers, planar re-layout on set / get_tensor, the pinned upload ring, the logits mirror
//   mi_planner.cpp        the fusion planner: graph walk, mul_mat / LayerNorm / attention chains, the plane pipeline, merged-chain walk
//   mi_batching.cpp       cross-state batching: rendezvous of decoding states, lanes, stream ordering
//   mi_distribution.cpp   one-time weight distribution to replicas (RCCL broadcast or peer copies) with device-side checksums
#pragma once
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

static inline bool env_flag(const char * name, bool def) {
    const char * v = getenv(name);
    if (!v || !*v) return def;
    return !(v[0] == '0' || v[0] == 'n' || v[0] == 'N' || v[0] == 'f' || v[0] == 'F');
}
static inline bool g_debug() { static bool d = env_flag("GGML_MI355X_DEBUG", false); return d; }
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
extern std::mutex                 g_weights_mtx;
extern std::vector<mi_weight_rec> g_buffers;
extern ggml_backend_reg           g_reg;
extern ggml_backend_device        g_devices[MI_MAX_DEVICES];
extern mi_device_ctx              g_device_ctx[MI_MAX_DEVICES];
extern int                        g_n_devices;
extern thread_local int       t_defer_weights;
extern std::atomic<uint64_t>  g_deferred_bytes;
// f16 copies of quantized WEIGHTS tensors that meet wide activations (encoder, cross-attention K/V, prompt): made once by
// mi355x_dequant_f16 the first time such a tensor reaches the MFMA path, kept until its buffer is written or freed.
// 2 bytes/weight of HBM buys a GEMM inner loop without dequantization (GGML_MI355X_F16_SHADOW_MB caps the total, 0 = off).
struct mi_shadow { void * f16; size_t bytes; const void * buf_base; int device; int type; int64_t ne0, ne1; };
extern std::mutex                                   g_shadow_mtx;
extern std::unordered_map<const void *, mi_shadow>  g_shadows;
extern std::atomic<size_t>                          g_shadow_count;
extern size_t                                       g_shadow_bytes;
extern std::atomic<uint64_t> g_io_ns[4];
extern std::atomic<uint64_t> g_io_calls[4];
struct io_timer {
    int slot; std::chrono::steady_clock::time_point t0;
    explicit io_timer(int s) : slot(s), t0(std::chrono::steady_clock::now()) {}
    ~io_timer() { g_io_ns[slot] += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_io_calls[slot]++; }
};
extern const bool g_trace;
extern std::atomic<uint64_t> g_trace_ns[8];
extern std::atomic<uint64_t> g_trace_calls[8];
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
extern mi_io_ctx g_io[MI_MAX_DEVICES];
#define MI_IO_SMALL (256u << 10)
#define MI_IO_DEFER (32u << 10)            // uploads up to this size wait in the ring for the next flush (graph inputs of a decode step)
// order `cs` behind every input upload accepted so far: deferred uploads (a step's graph inputs) leave with one scatter launch at the
// head of this stream; uploads that another stream flushed, or that went through async copies, are ordered in front of it by their events
struct mi_io_marks;
#define MI_REQUIRE_WHOLE_QUANT(t, what) do { if (is_quant_type((t)->type) && !whole_quant_tensor(t)) \
    GGML_ABORT("ggml-mi355x: %s of a partial / non-contiguous view of quantized tensor '%s' (%s): quantized tensors are stored planar and move as whole tensors only", what, (t)->name, ggml_type_name((t)->type)); } while (0)
extern const ggml_backend_buffer_i mi_buffer_iface;
extern const ggml_backend_buffer_type_i mi_buft_iface;
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
    std::atomic<bool> in_group{false};                          // counted among the device's decoding states: written under the group's mutex, READ without it by
                                                                // mi_batch_leave's fast path (ThreadSanitizer, r05: another thread's window time-out clears it meanwhile)
    bool        in_flight = false;                              // a column of a chain that is being launched right now (group's mutex)
    bool        own_dirty = false;                              // work was launched on the own stream since the group's stream last waited for it
    hipEvent_t  own_ev = nullptr;
    hipEvent_t  batch_wait_sync = nullptr, batch_wait_stream = nullptr;   // completion of the last batch this state was a column of: synchronize() / the own stream still have to wait for it
    int         no_batch_nodes = 0;                             // graph size (n_nodes) that was found not to fit the batch walker
    uint64_t    sig_nodes = 0; const void * sig_w = nullptr;    // graph shape of THIS state that was last checked congruent with its group's (mi_compute_batch)
    // host-visible mirror of the logits: the vocabulary projection stores its result a second time into pinned, device-mapped host memory,
    // so whisper's read-back of the row(s) (ggml_backend_tensor_get, src/whisper.cpp:2957-2963) is a memcpy instead of a device-to-host copy
    char *      mirror_host = nullptr; char * mirror_dev = nullptr;
    std::atomic<const void *> mirror_src{nullptr}; std::atomic<size_t> mirror_bytes{0};   // device range [mirror_src, + mirror_bytes) is what the mirror holds.  Atomic: every
                                                                // stream's logits read scans ALL backends' ranges while a chain leader publishes another state's (ThreadSanitizer, r05)
    std::atomic<int> mirror_state{0};                           // 0 nothing, 1 launched (not yet synchronized), 2 valid
    // where the last decoder step of this state left its logits (device): ggml_backend_mi355x_argmax_last reduces a row there
    const float * logits_dev = nullptr; int logits_n = 0, logits_rows = 0;
    std::atomic<uint64_t> n_graph_compute{0};            // (read by other states' threads: the batching-off warning)
    double   t_eager_ms = 0;                                    // host time inside graph_compute
    uint64_t trace_gc_enter = 0;                                // GGML_MI355X_TRACE: entry time of the graph_compute still waiting for its synchronize
    // GPU-side span of every graph_compute (first launch .. last kernel done), from a ring of hipEvent pairs on the stream
    std::vector<std::pair<hipEvent_t, hipEvent_t>> span_ev;
    int      span_next = 0, span_pending = 0;
    double   t_gpu_span_ms = 0;
};
extern double g_total_gpu_span_ms;
static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
extern std::vector<mi_backend_ctx *> g_backends;
extern std::shared_mutex g_backends_rw;
extern thread_local mi_backend_ctx * t_last_backend;
#define MI_MIRROR_CAP ((size_t) 2 << 20)
extern const bool g_mirror_on;
extern uint64_t g_total_stats[4];
extern double   g_total_host_ms[4];
// ---- mul_mat chain:  mul_mat [-> add bias] [-> scale] [-> gelu] [-> add residual] [-> cpy to f16] ----
struct mm_chain {
    const ggml_tensor * mm = nullptr;
    const ggml_tensor * last = nullptr;     // tensor whose memory receives the result
    mi355x_epilogue ep{};
    int end = 0;                            // index of the last fused node
    int res_node = -1, res_slot = 0;        // the residual operand is src[res_slot] of node res_node (cross-state batches look it up per graph)
};
// ---- norm [-> mul w -> add b] ---------------------------------------------------------------------
struct ln_chain { const ggml_tensor * norm = nullptr, * last = nullptr; const float * w = nullptr, * b = nullptr; int end = 0; };
// TEST fault injection (negative control of the parity tests: a test that cannot fail proves nothing).  GGML_MI355X_TEST_FAULT=
// "xattn:<n>:<factor>" multiplies the output of the n-th cross-attention block of a decoder step (n = -1: all of them) — the
// (W_o . attention + bias) that is added to the residual stream, src/whisper.cpp:2703-2770 — by `factor`, for steps of up to 8 columns.
// Cross-attention = the FLASH_ATTN_EXT nodes without a mask (decoder self-attention carries one; encoder attention has > 8 columns and
// never passes here).  Unset: the factor is exactly 1 and nothing changes.
// "reject:<n>": the n-th merged launch chain of the process (cross-state batching, counted from 1) reports a kernel-side rejection half-way
// through its walk — the mid-chain path of mi_compute_batch (drain, every member repeats the step on its own chain, the shape stops batching).
struct mi_test_fault { int layer = -2; float factor = 1.0f; int reject_chain = 0; };
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
struct mi_batch_member { mi_backend_ctx * b; ggml_cgraph * g; int state; ggml_status status;           // state: 0 waiting, 1 being launched, 2 done
                         int sig_nodes; const void * sig_w; };                                          // which model the step belongs to: node count + the vocabulary projection's weights
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
extern mi_batch_group     g_batch[MI_MAX_DEVICES];
extern std::atomic<int>   g_batching;
extern const ggml_backend_i mi_backend_iface;
extern const ggml_backend_device_i mi_dev_iface;
extern ggml_mi355x_feature g_features[];

// ---- functions shared between the translation units ----
bool is_quant_type(ggml_type t);
void mi_shadows_drop(int device, const void * buf_base);
mi_io_ctx * mi_io(int device);
void mi_io_flush_locked(mi_io_ctx & io, hipStream_t stream);
void mi_io_drain(int device);
bool mi_io_upload(int device, void * dst, const void * src, size_t size);
bool mi_io_download(int device, void * dst, const void * src, size_t size);
void mi_buffer_free(ggml_backend_buffer_t buffer);
void * mi_buffer_get_base(ggml_backend_buffer_t buffer);
bool whole_quant_tensor(const ggml_tensor * t);
void mi_buffer_set_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, const void * data, size_t offset, size_t size);
void mi_buffer_get_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * tensor, void * data, size_t offset, size_t size);
void mi_buffer_memset_tensor(ggml_backend_buffer_t buffer, ggml_tensor * tensor, uint8_t value, size_t offset, size_t size);
void mi_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value);
bool mi_buffer_is_ours(ggml_backend_buffer_t buffer);
bool mi_buffer_cpy_tensor(ggml_backend_buffer_t buffer, const ggml_tensor * src, ggml_tensor * dst);
const char * mi_buft_get_name(ggml_backend_buffer_type_t buft);
ggml_backend_buffer_t mi_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size);
size_t mi_buft_get_alignment(ggml_backend_buffer_type_t);
size_t mi_buft_get_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * tensor);
bool   mi_buft_is_host(ggml_backend_buffer_type_t);
void mi_span_drain(mi_backend_ctx * b);
void mi_io_order_stream(int device, mi_io_marks & mk, hipStream_t cs);
// the deferred uploads that wait in the ring and whose destination lies in one of `n` address ranges (a merged chain's members' compute buffers), records + bytes:
// what mi_io_replay stages again when the chain is rejected half-way and its members repeat the step (their inputs' memory may have been reused by then)
struct mi_io_saved { std::vector<mi_io_rec> recs; std::vector<char> data; };
void mi_io_snapshot(int device, const char * const * base, const size_t * size, int n, mi_io_saved & out);
void mi_io_replay(int device, const mi_io_saved & s);
char * mi_mirror_dev(mi_backend_ctx * b);
void mi_mirror_invalidate(int device, const void * p, size_t n);
bool mi_mirror_read(int device, const void * src, void * dst, size_t size);
mi355x_tensor to_mt(const ggml_tensor * t);
bool op_is_empty(const ggml_tensor * t);
int use_count(const ggml_cgraph * g, const ggml_tensor * t);
bool can_elide(const ggml_cgraph * g, const ggml_tensor * t, int n);
bool overlap(const void * a, size_t na, const void * b, size_t nb);
bool t_overlap(const ggml_tensor * a, const ggml_tensor * b);
bool is_vec_f32(const ggml_tensor * t, int64_t n);
int next_real(const ggml_cgraph * g, int i);
bool parse_mm_chain(const ggml_cgraph * g, int i, bool fuse, mm_chain & c);
int mode_for(ggml_type t);
bool mi_mmq_on();
int mi_mmq_mode();
int rows_mode_for(const ggml_tensor * w, int64_t K, int64_t T);
const void * mi_shadow_get(mi_backend_ctx * b, const ggml_tensor * w, const mi355x_tensor & mw);
int mi_act_reserve(mi_backend_ctx * b, size_t need, bool alt = false);
bool mm_takes_prepared(const mi_backend_ctx * b, const ggml_tensor * mm, const ggml_tensor * x, int & mode_out);
int run_mm_chain(mi_backend_ctx * b, const mm_chain & c, const ggml_cgraph * g = nullptr);
void parse_ln_chain(const ggml_cgraph * g, int i, bool fuse, ln_chain & c);
int run_ln_chain(mi_backend_ctx * b, const ln_chain & c, const ggml_cgraph * g = nullptr);
const mi_test_fault & mi_fault();
float mi_fault_scale(const ggml_cgraph * g, int i);
void mi_fault_apply(const ggml_cgraph * g, int i, mi355x_epilogue & ep);
bool mirror_wanted(const ggml_cgraph * g, const mm_chain & ch, int64_t T);
bool try_ln_gemv(mi_backend_ctx * b, const ggml_cgraph * g, const ln_chain & ln, int & end_out, int & rc_out);
bool try_fattn_gemv(mi_backend_ctx * b, const ggml_cgraph * g, int i, int & end_out, int & rc_out);
void attn_consume(mi_backend_ctx * b, const ggml_cgraph * g, int i, const mi355x_attn_partials & parts, int & end_out, int & rc_out);
bool whole_quant_ok(const ggml_tensor * t);
bool mi_supports_op_impl(const ggml_tensor * op);
int run_node(mi_backend_ctx * b, const ggml_tensor * n);
const ggml_tensor * cs_tensor(const mi_colset & cs, int c, int node, int slot);
char * cs_col(const mi_colset & cs, int c, int node, int slot, int64_t nb1 = -1);
bool q_weight_ok(const ggml_tensor * w, int64_t K);
void q_fill_seg(const mi_colset & cs, const mm_chain & ch, int s, mi355x_gemv_desc & d, mi355x_gemv_cols & cols);
bool q_ln_gemv(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, const ln_chain & ln, int & end_out, int & rc_out);
bool q_attn_proj(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, int & end_out, int & rc_out);
bool q_mm(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, int & end_out, int & rc_out);
int mi_walk_batch(mi355x_ctx * k, const mi_colset & cs);
bool mi_graphs_congruent(const ggml_cgraph * a, const ggml_cgraph * b);
int mi_emit_range(mi_backend_ctx * b, ggml_cgraph * g, int i0, int i_stop);
int mi_emit_graph(mi_backend_ctx * b, ggml_cgraph * g);
const char * mi_backend_get_name(ggml_backend_t backend);
void mi_backend_free(ggml_backend_t backend);
void mi_backend_synchronize(ggml_backend_t backend);
ggml_status mi_compute_own(mi_backend_ctx * b, ggml_cgraph * cgraph);
bool mi_batching_on();
int mi_batch_min_states();
bool mi_is_step_graph(const ggml_cgraph * g);
void mi_batch_leave(mi_backend_ctx * b);
bool mi_compute_batch(mi_batch_group & grp, mi_batch_group::lane & ln, mi_batch_member ** mem, int n);
ggml_status mi_batch_join(mi_backend_ctx * b, ggml_cgraph * cgraph);
ggml_status mi_backend_graph_compute(ggml_backend_t backend, ggml_cgraph * cgraph);
ggml_guid_t mi_guid();
const char * mi_dev_get_name(ggml_backend_dev_t dev);
const char * mi_dev_get_description(ggml_backend_dev_t dev);
void mi_dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total);
enum ggml_backend_dev_type mi_dev_get_type(ggml_backend_dev_t);
void mi_dev_get_props(ggml_backend_dev_t dev, ggml_backend_dev_props * props);
ggml_backend_t mi_dev_init_backend(ggml_backend_dev_t dev, const char *);
ggml_backend_buffer_type_t mi_dev_get_buffer_type(ggml_backend_dev_t dev);
bool mi_dev_supports_op(ggml_backend_dev_t, const ggml_tensor * op);
bool mi_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft);
void mi_init_devices();
const char * mi_reg_get_name(ggml_backend_reg_t);
size_t mi_reg_get_device_count(ggml_backend_reg_t);
ggml_backend_dev_t mi_reg_get_device(ggml_backend_reg_t, size_t index);
mi_backend_ctx * as_ctx(void * backend);
