// whisper_lang_auto_detect (src/whisper.cpp:4047-4121 — one encode, one decode of the SOT token, softmax
// over the language tokens) through the MI355X plugin against the reference CPU backend, same multilingual model file and mel.
// Prints one JSON object: the detected language on both sides, the probability vectors' largest difference.  TEST code.
#include "whisper.h"
#include "ggml-backend.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static void log_quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s multilingual-model.bin\n  env GGML_MI355X_PLUGIN=path\n", argv[0]); return 2; }
    whisper_log_set(log_quiet, nullptr);
    const char * plugin = getenv("GGML_MI355X_PLUGIN");
    const bool selftest = plugin && !strcmp(plugin, "cpu");
    if (!selftest && (!plugin || !ggml_backend_load(plugin))) { fprintf(stderr, "cannot load plugin (GGML_MI355X_PLUGIN)\n"); return 3; }
    const int n_threads = 8, n_lang = whisper_lang_max_id() + 1;
    std::vector<float> probs[2];
    int best[2] = { -1, -1 };
    for (int side = 0; side < 2; side++) {
        whisper_context_params cp = whisper_context_default_params();
        cp.flash_attn = true; cp.use_gpu = side == 1 && !selftest; cp.gpu_device = 0;
        whisper_context * ctx = whisper_init_from_file_with_params(argv[1], cp);
        if (!ctx) { fprintf(stderr, "model load failed\n"); return 3; }
        if (!whisper_is_multilingual(ctx)) { fprintf(stderr, "model is not multilingual\n"); return 3; }
        const int n_mels = whisper_model_n_mels(ctx), n_len = 3000;
        std::vector<float> mel((size_t) n_mels * n_len);
        std::mt19937 rng(7);
        for (int j = 0; j < n_mels; j++) for (int i = 0; i < n_len; i++)
            mel[(size_t) j * n_len + i] = 0.5f * sinf(0.011f * i + 0.17f * j) + 0.5f * ((rng() >> 8) * (1.0f / 8388608.0f) - 1.0f);
        whisper_set_mel(ctx, mel.data(), n_len, n_mels);
        probs[side].assign(n_lang, 0.0f);
        best[side] = whisper_lang_auto_detect(ctx, 0, n_threads, probs[side].data());
        if (best[side] < 0) { fprintf(stderr, "whisper_lang_auto_detect failed: %d\n", best[side]); return 4; }
        whisper_free(ctx);
    }
    double max_diff = 0, sum0 = 0, sum1 = 0, pmax = 0, second = 0;
    for (int i = 0; i < n_lang; i++) {
        max_diff = std::max(max_diff, (double) fabsf(probs[0][i] - probs[1][i])); sum0 += probs[0][i]; sum1 += probs[1][i];
        if (probs[0][i] > pmax) { second = pmax; pmax = probs[0][i]; } else if (probs[0][i] > second) second = probs[0][i];
    }
    printf("{\"model\": \"%s\", \"n_lang\": %d, \"lang_cpu\": \"%s\", \"lang_gpu\": \"%s\", \"id_cpu\": %d, \"id_gpu\": %d, \"p_top_cpu\": %.6f, \"p_second_cpu\": %.6f, "
           "\"max_abs_prob_diff\": %.3e, \"sum_cpu\": %.6f, \"sum_gpu\": %.6f}\n", argv[1], n_lang, whisper_lang_str(best[0]), whisper_lang_str(best[1]), best[0], best[1],
           pmax, second, max_diff, sum0, sum1);
    return 0;
}
