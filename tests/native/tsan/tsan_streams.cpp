// TEST driver (never shipped): S whisper_states on S host threads through the unmodified libwhisper and the plugin built with
// -fsanitize=thread on top of the stub device (hip_stub.cpp, kernels_stub.cpp): encode + n_decode single-token decodes per chunk, cross-state
// batching on from 2 states, chunk boundaries staggered so that states join and leave the rendezvous while chains are in flight.
// ThreadSanitizer watches the plugin's host logic (mi_batching.cpp, mi_buffers.cpp: rendezvous, lanes, upload ring, logits mirror).
//   usage: tsan_streams <model> <plugin.so> [streams] [chunks] [n_decode]
#include "whisper.h"
#include "ggml-backend.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

int main(int argc, char ** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s model plugin [streams] [chunks] [n_decode]\n", argv[0]); return 2; }
    const int S = argc > 3 ? atoi(argv[3]) : 6, chunks = argc > 4 ? atoi(argv[4]) : 3, n_decode = argc > 5 ? atoi(argv[5]) : 24;
    ggml_backend_reg_t reg = ggml_backend_load(argv[2]);
    if (!reg) { fprintf(stderr, "ggml_backend_load(%s) failed\n", argv[2]); return 3; }
    typedef void (*set_batching_t)(int);
    set_batching_t set_batching = (set_batching_t) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_set_batching");
    if (set_batching) set_batching(2);                       // merged chains from two decoding states on
    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = true; cp.flash_attn = true; cp.gpu_device = 0;
    whisper_context * ctx = whisper_init_from_file_with_params_no_state(argv[1], cp);
    if (!ctx) { fprintf(stderr, "model load failed\n"); return 4; }
    const int n_mels = whisper_model_n_mels(ctx);
    std::vector<whisper_state *> st(S);
    for (int s = 0; s < S; s++) { st[s] = whisper_init_state(ctx); if (!st[s]) { fprintf(stderr, "whisper_init_state failed\n"); return 5; } }
    std::vector<int> rc(S, 0);
    std::vector<std::thread> th;
    for (int s = 0; s < S; s++) th.emplace_back([&, s] {
        std::vector<float> mel((size_t) n_mels * 3000, 0.01f * (s + 1));
        if (whisper_set_mel_with_state(ctx, st[s], mel.data(), 3000, n_mels) != 0) { rc[s] = 10; return; }
        whisper_token tok[1] = { 0 };
        for (int c = 0; c < chunks && !rc[s]; c++) {
            if (whisper_encode_with_state(ctx, st[s], 0, 2) != 0) { rc[s] = 11; return; }
            const int nd = n_decode + 3 * s;             // staggered chunk boundaries
            for (int i = 0; i < nd; i++) if (whisper_decode_with_state(ctx, st[s], tok, 1, i, 2) != 0) { rc[s] = 12; return; }
            volatile float sink = whisper_get_logits_from_state(st[s])[0]; (void) sink;
        }
    });
    for (auto & t : th) t.join();
    int bad = 0;
    for (int s = 0; s < S; s++) if (rc[s]) { fprintf(stderr, "stream %d failed: %d\n", s, rc[s]); bad = 1; }
    for (int s = 0; s < S; s++) whisper_free_state(st[s]);
    whisper_free(ctx);
    // merged launch chains, columns they carried, steps a state ran alone, fall-backs, closed windows (include/ggml_mi355x.h)
    typedef void (*batch_stats_t)(int, uint64_t *);
    batch_stats_t batch_stats = (batch_stats_t) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_batch_stats");
    uint64_t bs[5] = { 0, 0, 0, 0, 0 };
    if (batch_stats) batch_stats(0, bs);
    printf("tsan_streams: %d streams x %d chunks done%s chains=%llu columns=%llu solo=%llu fallbacks=%llu timeouts=%llu\n", S, chunks, bad ? " (with failures)" : "",
           (unsigned long long) bs[0], (unsigned long long) bs[1], (unsigned long long) bs[2], (unsigned long long) bs[3], (unsigned long long) bs[4]);
    return bad;
}
