/*
 * mi355x_host.h — native host harness above whisper.h for the MI355X plugin (C ABI, libmi355x_host.so).
 *
 * whisper.cpp's own way of running several streams is one host thread per whisper_state (whisper_full_parallel,
 * src/whisper.cpp:7848-7869) and one whisper_context per device (whisper_context_params.gpu_device, include/whisper.h:119).
 * This harness is that arrangement for the benchmark protocol (examples/bench/bench.cpp:124-136: 1 x whisper_encode +
 * n_decode x whisper_decode per 30 s chunk), in C++ threads instead of Python threads:
 *     n_devices contexts (device r for context r), streams_per_device whisper_states on each, one thread per state.
 * Weights reach device r > 0 without touching the model file's tensor payloads: a whisper_model_loader (include/whisper.h:153-159)
 * that serves the header / filters / vocabulary and reports end-of-file where the tensors begin (the reference then allocates the
 * tensors and loads none: src/whisper.cpp:1944-1950), followed by the plugin's device-to-device broadcast from device 0 and a
 * checksum comparison of every weights buffer (include/ggml_mi355x.h).  A broadcast that cannot be verified is an error.
 * With use_gpu = 0 the same harness runs on the reference CPU backend (every context reads the whole file): that is how the
 * threading / timing logic is covered on a machine without a GPU.
 */
#ifndef MI355X_HOST_H
#define MI355X_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_HOST_API __attribute__((visibility("default")))

typedef struct mi355x_host_config {
    const char * model_path;
    const char * plugin_path;      /* libggml-mi355x.so; NULL = already loaded / CPU */
    int32_t use_gpu;               /* 1: MI355X plugin, 0: reference CPU backend */
    int32_t n_devices;             /* contexts; context r uses gpu_device = first_device + r */
    int32_t first_device;
    int32_t streams_per_device;    /* whisper_states per context */
    int32_t n_decode;              /* single-token decodes per chunk (bench protocol: 256) */
    int32_t steps, warmup;         /* chunks per stream: timed, untimed */
    int32_t n_threads;             /* n_threads argument of whisper_encode / whisper_decode */
    int32_t skip_payloads;         /* 1: contexts r > 0 load through the payload-skipping loader + broadcast (GPU only) */
    int32_t flash_attn;
    int32_t replicas_on_one_device; /* 1: every context uses gpu_device = first_device (a one-GPU machine standing in for n_devices GPUs:
                                     *    the payload-skipping load + device copy + checksum verify path runs on real hardware) */
    int32_t device_greedy;         /* 1 (GPU only): every stream decodes FREE-RUNNING — the token fed to step i + 1 is the arg-max of step i's logits, taken ON
                                    * THE DEVICE by ggml_backend_mi355x_argmax_last (16 bytes back instead of a host scan of n_vocab floats) and checked
                                    * against the host scan of the row whisper_decode returned (result: greedy_checked / greedy_mismatches) */
    int32_t batching;              /* cross-state batching in the plugin (ggml_backend_mi355x_set_batching): 1 on (merged chains from 5 states on), n >= 2 on from n states, 0 off, -1 leave as it is.  With
                                    * it the states of one device that decode at the same time run as the columns of ONE launch chain */
    int32_t transport;             /* how the weights reach contexts r > 0 (skip_payloads = 1): 0 = RCCL — one communicator per device in this process, one grouped
                                    * ncclBroadcast per weights buffer from first_device (ggml_backend_mi355x_broadcast_weights_rccl_group; the default, and an error
                                    * if librccl cannot be loaded: no silent substitute); 1 = hipMemcpyPeerAsync device 0 -> device r (ggml_backend_mi355x_broadcast_
                                    * weights_peer); 2 = RCCL also with n_devices = 1: the world-1 communicator, broadcast and checksum verification run
                                    * (what a one-GPU box can show of the path; with 0 a single context distributes nothing).  replicas_on_one_device = 1 always uses the on-device copy (RCCL admits one rank per device) */
} mi355x_host_config;

typedef struct mi355x_host_result {
    double  wall_s;                /* timed region, all threads started together */
    double  chunks_per_s;          /* (n_devices * streams_per_device * steps) / wall_s */
    double  ms_per_chunk_per_stream;
    double  load_s;                /* all contexts + states */
    double  bcast_bytes, bcast_seconds;
    int32_t bcast_buffers, bcast_verified;     /* verified: every destination buffer's checksum equals device 0's */
    int32_t bcast_transport, bcast_ranks;      /* what moved the weights: 0 nothing (one context, or every context read the file), 1 RCCL (ranks = communicator size), 2 peer copies, 3 copy on one device */
    double  bcast_setup_seconds;               /* RCCL: ncclCommInitAll */
    double  encode_ms, decode_ms_per_token;    /* mean over all streams and timed chunks: wall milliseconds of whisper_encode, of one whisper_decode */
    int64_t payload_bytes_read;    /* bytes the loaders actually read from the model file, summed over contexts */
    int64_t file_bytes;
    int32_t n_devices, streams_per_device;
    char    error[256];            /* empty on success */
    int64_t greedy_checked, greedy_mismatches;   /* device_greedy: tokens compared with the host arg-max of the same logits row, and how many differed */
    uint64_t batch_stats[5];       /* device first_device, over the whole run: merged launch chains, columns they carried, steps a state ran alone,
                                    * groups that fell back to one chain per state, windows closed on an absent state (ggml_backend_mi355x_batch_stats) */
} mi355x_host_result;

/* returns 0 on success; on failure a non-zero code and out->error */
MI355X_HOST_API int mi355x_host_run(const mi355x_host_config * cfg, mi355x_host_result * out);

/* the payload-skipping loader on its own (tests): opens the model with it on the CPU backend, frees it again.
 * out[0] = bytes read through the loader, out[1] = file size, out[2] = byte offset at which the tensors begin. */
MI355X_HOST_API int mi355x_host_probe_skipping_loader(const char * model_path, int64_t * out3);

/* One whisper_context WITH its default state for hosts that drive whisper.h themselves (one process per device): skip_payloads = 1
 * opens the file through the payload-skipping loader; the caller then fills the weights with the plugin's broadcast before the first
 * whisper_encode.  Returns struct whisper_context * (NULL on failure); *bytes_read = bytes actually read from the model file. */
MI355X_HOST_API void * mi355x_host_open(const char * model_path, int use_gpu, int gpu_device, int flash_attn, int skip_payloads, int64_t * bytes_read);

/* rows of every stream's last logits (n_vocab floats each) after a run, for bit-identity checks between arrangements */
MI355X_HOST_API int mi355x_host_last_logits(float * dst, int64_t cap_floats);
/* one step of the whisper-bench protocol (examples/bench/bench.cpp:124-136) on an open whisper_context: whisper_encode + n_decode x
 * whisper_decode(1 token, n_past = i); 0, or 1 / 2 if the encode / a decode failed.  What bench.py times (no host-language call per token). */
MI355X_HOST_API int mi355x_host_chunk(void * whisper_ctx, int n_decode, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
