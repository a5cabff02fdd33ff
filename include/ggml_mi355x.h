/*
 * ggml_mi355x.h — the drop-in boundary: a ggml backend plugin for AMD Instinct MI355X (gfx950).
 *
 * `libggml-mi355x.so` is loaded by an UNMODIFIED whisper.cpp / ggml through ggml's own plugin loader:
 *     GGML_BACKEND_PATH=/path/to/libggml-mi355x.so whisper-bench -m model.bin
 * (ggml/src/ggml-backend-reg.cpp:562-593 `ggml_backend_load_all`, :220-264 `load_backend`, which dlsym()s the two
 * C symbols below).  Everything else crosses the boundary through the C function-pointer tables of
 * ggml/src/ggml-backend-impl.h (api_version GGML_BACKEND_API_VERSION == 2):
 *     ggml_backend_reg_i          :214-224   get_name, get_device_count, get_device, get_proc_address
 *     ggml_backend_device_i       :160-202   get_type -> GPU, init_backend, get_buffer_type, supports_op, supports_buft ...
 *     ggml_backend_buffer_type_i  :17-29     alloc_buffer (hipMalloc), get_alignment, get_alloc_size
 *     ggml_backend_buffer_i       :41-62     get_base, set_tensor / get_tensor (planar re-layout of quantized
 *                                            weights happens here), cpy_tensor, clear, memset_tensor
 *     ggml_backend_i              :105-140   graph_compute (fusion planner + HIP kernels), synchronize
 * No reference source file or build flag changes; whisper.h / whisper_full() are untouched.
 *
 * The declarations below use `void *` where ggml uses its opaque handle typedefs so that this header can be
 * included without ggml's headers.
 */
#ifndef GGML_MI355X_H
#define GGML_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MI355X_API __attribute__((visibility("default")))

/* ---- the two symbols ggml's loader binds (ggml-backend-impl.h:229-235) --------------------------- */

/* returns ggml_backend_reg_t (struct ggml_backend_reg *) with api_version == 2; NULL if no gfx950 device */
GGML_MI355X_API void * ggml_backend_init(void);

/* 0 = "not usable on this system" (no gfx950 GPU), otherwise a positive score */
GGML_MI355X_API int    ggml_backend_score(void);

/* ---- backend-private entry points, also reachable via ggml_backend_reg_get_proc_address(reg, name) --- */

/* ggml_backend_reg_t for explicit (non-dlopen) registration: ggml_backend_register(ggml_backend_mi355x_reg()) */
GGML_MI355X_API void * ggml_backend_mi355x_reg(void);

/* "ggml_backend_get_features": array of {name, value} string pairs terminated by {NULL, NULL}
 * (printed by whisper_print_system_info, src/whisper.cpp:4351) */
struct ggml_mi355x_feature { const char * name; const char * value; };
GGML_MI355X_API struct ggml_mi355x_feature * ggml_backend_mi355x_get_features(void * reg);

/* "ggml_backend_set_n_threads" (looked up by name by src/whisper.cpp:191-205 before every graph): the HIP path has no threads;
 * the value only selects the key-range chunking of the reference-exact attention mode (GGML_MI355X_EXACT=1), because the
 * reference CPU flash attention splits single-query steps over its threads (ggml-cpu/ops.cpp:9117-9150) */
GGML_MI355X_API void ggml_backend_mi355x_set_n_threads(void * backend, int n_threads);

/* Per-kernel profile of one backend (ggml_backend_t): enables hipEvent bracketing of every launch on the
 * backend's stream.  Rows as in mi355x_prof_row. */
struct ggml_mi355x_prof_row { const char * name; uint64_t calls; double total_ms; double algo_bytes; double algo_flops; };
GGML_MI355X_API void ggml_backend_mi355x_prof_enable(void * backend, int on);
GGML_MI355X_API void ggml_backend_mi355x_prof_reset(void * backend);
GGML_MI355X_API int  ggml_backend_mi355x_prof_report(void * backend, struct ggml_mi355x_prof_row * rows, int cap);

/* process-wide variants over every live MI355X backend (whisper.h does not expose its ggml_backend_t handles);
 * ggml_backend_mi355x_stats fills out[4] = {graph_compute calls, 0, 0, 0} (slots 1..3 belonged to the removed hipGraph replay path) */
GGML_MI355X_API void ggml_backend_mi355x_prof_enable_all(int on);
GGML_MI355X_API void ggml_backend_mi355x_prof_reset_all(void);
GGML_MI355X_API int  ggml_backend_mi355x_prof_report_all(struct ggml_mi355x_prof_row * rows, int cap);
GGML_MI355X_API void ggml_backend_mi355x_stats(uint64_t * out);
/* out[13]: [12] GPU-side span of all completed graph_computes (ms, hipEvent pairs on the compute stream); [3] host milliseconds inside graph_compute (graph walk + launches; [0..2] always 0, kept for layout); [4..7] milliseconds inside {set_tensor, get_tensor, cpy_tensor, synchronize}; [8..11] their call counts */
GGML_MI355X_API void ggml_backend_mi355x_host_times(double * out);
/* GGML_MI355X_TRACE=1 (off: returns 0, zeros): out[2i] = nanoseconds, out[2i+1] = calls of slot i — 0 supports_op, 1 supports_buft,
 * 2 graph_compute entry -> first kernel launched, 3 graph_compute entry -> return, 4 graph_compute entry -> synchronize return,
 * 5 get_proc_address.  The plugin-owned side of the per-step host timeline (tests/native/step_trace.cpp measures the reference's side). */
GGML_MI355X_API int  ggml_backend_mi355x_trace(uint64_t * out16);

/* Cross-state batching: whisper_states of one device whose next graph is a single-token decoder step are
 * executed as the COLUMNS of one launch chain — every weight byte is read once for all of them instead of once per state — by whichever
 * of their host threads completes the set (up to 32 columns per chain, MI355X_IMG_COLS x 4; a state alone runs the ordinary path; a state whose next graph is anything
 * else leaves the group at once).  Per state the results are bit-identical to running alone.  ON by default since round 4 (a
 * whisper_full_parallel user with 8 states otherwise gets the own-chain collapse): GGML_MI355X_BATCH=0 or
 * ggml_backend_mi355x_set_batching(0) switch it off at any time; ggml_backend_mi355x_get_batching returns the current setting.  With
 * fewer than 5 decoding states (GGML_MI355X_BATCH_MIN_STREAMS) every state still runs its own chain — measured faster on one MI355X up
 * to 4 states; an argument n >= 2 (or GGML_MI355X_BATCH=n) merges from n states on.
 * ggml_backend_mi355x_batch_stats: out[0..4] = merged chains, columns carried, steps
 * run alone, groups that fell back to one chain per state, windows that closed on an absent state. */
GGML_MI355X_API void ggml_backend_mi355x_set_batching(int on);
GGML_MI355X_API int  ggml_backend_mi355x_get_batching(void);
/* Device-side greedy sampling: most probable token of logits row `row` (-1: last) of the decoder step the CALLING
 * THREAD issued last, reduced in HBM (16 bytes come back); *top1 = its logit, *margin = distance to the runner-up.  -1: no such step.
 * For hosts with their own decoding loop above whisper_decode (include/mi355x_host.h greedy mode); whisper_full's sampler is untouched. */
GGML_MI355X_API int  ggml_backend_mi355x_argmax_last(int row, float * top1, float * margin);
GGML_MI355X_API void ggml_backend_mi355x_batch_stats(int device, uint64_t * out5);
/* test hook, needs no device (nothing is launched): does `cgraph` (struct ggml_cgraph *) fit the cross-state walker as S >= 2 columns
 * (out[0] == 0), resp. for S == 1 how many of its stages the T >= 3 plane pipeline takes (out[1..3] LayerNorm / attention / plain
 * mat-vec stages, out[4] compute nodes left to the other paths) */
GGML_MI355X_API int  ggml_backend_mi355x_debug_walk(void * cgraph, int S, int64_t * out6);

/* Multi-GPU weight distribution.  Replicas are independent streams: the only exchange is the ONE-TIME copy of
 * rank 0's WEIGHTS buffers into the identically laid out buffers of the other replicas (every context allocates the same tensors in
 * the same order, src/whisper.cpp:1685-1859; buffer counts and sizes are compared first).  The reference has no such call site: it
 * re-reads the model file and uploads tensor by tensor for every context (src/whisper.cpp:1934-1938).
 *   ggml_backend_mi355x_broadcast_weights_rccl_group  one process, n devices (what `bench.py --gpus N` and mi355x_host_run use): one RCCL
 *                                                communicator per device (ncclCommInitAll) and ONE grouped ncclBroadcast per weights buffer from
 *                                                devices[0] (the reference's in-process pattern: ggml-cuda/ggml-cuda.cu:1188, :1027-1030), then
 *                                                checksums; stats6 = the four below + seconds of communicator setup + ranks; n == 1 is valid
 *   ggml_backend_mi355x_broadcast_weights_peer   one process, two devices: hipMemcpyPeerAsync over xGMI, then checksums (only on request:
 *                                                mi355x_host_config.transport = 1, bench.py --transport peer)
 *   ggml_backend_mi355x_clone_weights            n contexts created one after the other on ONE device (a one-GPU machine standing in for
 *                                                n GPUs in tests): copy the first context's buffers into the others', verify the same way
 *   ggml_backend_mi355x_rccl_unique_id           rank 0 of a multi-process job: 128-byte id for the host harness to hand to every rank
 *   ggml_backend_mi355x_broadcast_weights_rccl   one process per device: RCCL communicator from that id (librccl.so is dlopen()ed),
 *                                                ncclBroadcast of every buffer from rank 0, device-side checksums compared across ranks
 *   ggml_backend_mi355x_weights_checksum         {sum, index-weighted sum} mod 2^64 of every WEIGHTS buffer (mi355x_checksum)
 *   ggml_backend_mi355x_defer_weights            while on, set_tensor of weight tensors issued BY THE CALLING THREAD is skipped (they arrive by
 *                                                broadcast); ggml_backend_mi355x_deferred_bytes = bytes skipped so far, process-wide
 *   ggml_backend_mi355x_weight_buffers           base pointers + sizes in allocation order (for harnesses with their own transport)
 * stats[0..3] = bytes moved, seconds, buffers, verified (1: every destination buffer's checksum equals the source's).  Return 0 on
 * success; -1 transport unavailable / failed, -2 layout differs between replicas, -3 copy failed, -4 checksum mismatch. */
GGML_MI355X_API int  ggml_backend_mi355x_weight_buffers(int device, void ** bases, size_t * sizes, int cap);
GGML_MI355X_API int  ggml_backend_mi355x_weights_checksum(int device, uint64_t * out_pairs, int cap);
GGML_MI355X_API int  ggml_backend_mi355x_broadcast_weights_peer(int src_device, int dst_device, double * stats4);
GGML_MI355X_API int  ggml_backend_mi355x_broadcast_weights_rccl_group(const int * devices, int n, double * stats6);
GGML_MI355X_API int  ggml_backend_mi355x_clone_weights(int device, int n_replicas, double * stats4);
GGML_MI355X_API int  ggml_backend_mi355x_rccl_unique_id(void * out128);
GGML_MI355X_API int  ggml_backend_mi355x_broadcast_weights_rccl(int device, int rank, int world, const void * unique_id128, double * stats4);
GGML_MI355X_API void ggml_backend_mi355x_defer_weights(int on);
GGML_MI355X_API uint64_t ggml_backend_mi355x_deferred_bytes(void);

/* Runtime switches (environment):
 *   GGML_MI355X_FUSE=0       run every ggml node as its own kernel (debug / parity bisect)
 *   GGML_MI355X_DEBUG=1      log unsupported ops and kernel-library errors to stderr
 *   GGML_MI355X_STRICT=1     abort instead of letting the scheduler fall back to the CPU backend for an unsupported op
 *   GGML_MI355X_EXACT=1      reference-exact arithmetic (test mode, slow): flash attention as the CPU dispatcher computes it
 *                            (F16 accumulation / split over n_threads / F32 tiles), integer block dots for every column count
 * (the full list of the environment switches: INTEGRATION.md section 6; kernel-shape variants that tests compare are not
 *  environment switches but mi355x_test_option values, include/mi355x_kernels.h)
 * The plugin sets GPU_MAX_HW_QUEUES=8 when it is loaded unless the variable is already set (one hardware queue per concurrent stream).
 */

#ifdef __cplusplus
}
#endif
#endif /* GGML_MI355X_H */
