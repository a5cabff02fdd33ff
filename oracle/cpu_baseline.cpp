// CPU baseline leg of bench.py: the reference's own AVX2 CPU path (oracle/_ref/libwhisper.so, use_gpu = false)
// timed on a BOUNDED sample of the whisper-bench protocol (examples/bench/bench.cpp:63-170):
//   1 warm-up + 1 timed whisper_encode, `n_decode` timed single-token whisper_decode steps (n_past = i).
// Prints one JSON object.  TEST / BASELINE infrastructure — reported next to the GPU number, never the target.
#include "whisper.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static void quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

int main(int argc, char ** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model.bin n_threads n_decode [warm_encode=1]\n", argv[0]); return 2; }
    const char * model = argv[1];
    const int n_threads = atoi(argv[2]), n_decode = atoi(argv[3]);
    const int warm = argc > 4 ? atoi(argv[4]) : 1;
    whisper_log_set(quiet, nullptr);
    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = false; cp.flash_attn = true;
    whisper_context * ctx = whisper_init_from_file_with_params(model, cp);
    if (!ctx) return 3;
    const int n_mels = whisper_model_n_mels(ctx), n_len = 3000;
    std::vector<float> mel((size_t) n_mels * n_len);
    std::mt19937 rng(42);
    for (auto & v : mel) v = (rng() >> 8) * (2.0f / 16777216.0f) - 1.0f;
    whisper_set_mel(ctx, mel.data(), n_len, n_mels);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (warm && whisper_encode(ctx, 0, n_threads) != 0) return 4;
    auto t0 = now();
    if (whisper_encode(ctx, 0, n_threads) != 0) return 4;
    auto t1 = now();
    whisper_token tok = 0;
    if (whisper_decode(ctx, &tok, 1, 0, n_threads) != 0) return 4;       // warm the decoder graph
    auto t2 = now();
    for (int i = 0; i < n_decode; i++) if (whisper_decode(ctx, &tok, 1, i, n_threads) != 0) return 4;
    auto t3 = now();
    printf("{\"encode_ms\": %.3f, \"decode_ms_per_token\": %.4f, \"n_decode\": %d, \"threads\": %d, \"system_info\": \"%s\"}\n",
           ms(t0, t1), ms(t2, t3) / n_decode, n_decode, n_threads, whisper_print_system_info());
    whisper_free(ctx);
    return 0;
}
