// Is a decoder step deterministic?  The same whisper_decode call (same tokens, same position, same KV prefix) is issued `reps` times
// through the plugin and its logits are compared BIT FOR BIT with the first run's; the reference CPU backend is not involved, so a
// difference can only be a race / an uninitialised read in the plugin's launch chain.  (Round 3: the closing run's 5-token batch of
// two models was off by 2e-2 .. 2e-1 NMSE in some runs and exact in others.)
//
//   repeat_check model.bin [n_tokens=5] [reps=40] [flash_attn=1] [n_past=3]     env: GGML_MI355X_PLUGIN
//
// Output: one JSON object {n_tokens, reps, mismatching_runs, first_bad_rep, max_abs_diff, rows_touched}.  TEST code.
#include "whisper.h"
#include "ggml.h"
#include "ggml-backend.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static void log_quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.bin [n_tokens=5] [reps=40] [flash_attn=1] [n_past=3]\n", argv[0]); return 2; }
    const int n_tok = argc > 2 ? atoi(argv[2]) : 5, reps = argc > 3 ? atoi(argv[3]) : 40, fa = argc > 4 ? atoi(argv[4]) : 1, n_past = argc > 5 ? atoi(argv[5]) : 3;
    whisper_log_set(log_quiet, nullptr);
    ggml_log_set(log_quiet, nullptr);
    const char * plugin = getenv("GGML_MI355X_PLUGIN");
    if (!plugin || !ggml_backend_load(plugin)) { fprintf(stderr, "cannot load plugin (GGML_MI355X_PLUGIN)\n"); return 3; }
    whisper_context_params cp = whisper_context_default_params();
    cp.flash_attn = fa != 0; cp.use_gpu = true; cp.gpu_device = 0;
    whisper_context * ctx = whisper_init_from_file_with_params(argv[1], cp);
    if (!ctx) { fprintf(stderr, "model load failed\n"); return 3; }
    const int n_mels = whisper_model_n_mels(ctx), n_len = 3000, n_vocab = whisper_n_vocab(ctx);
    std::vector<float> mel((size_t) n_mels * n_len);
    std::mt19937 rng(42);
    for (auto & x : mel) x = (rng() >> 8) * (2.0f / 16777216.0f) - 1.0f;
    whisper_set_mel(ctx, mel.data(), n_len, n_mels);
    if (whisper_encode(ctx, 0, 4) != 0) { fprintf(stderr, "encode failed\n"); return 4; }
    std::vector<whisper_token> prefix(n_past > 0 ? n_past : 1), tok(n_tok);
    for (auto & t : prefix) t = (whisper_token) (rng() % (n_vocab - 2000));
    for (auto & t : tok)    t = (whisper_token) (rng() % (n_vocab - 2000));
    for (int i = 0; i < n_past; i++) if (whisper_decode(ctx, &prefix[i], 1, i, 4) != 0) return 4;       // a KV prefix, token by token
    std::vector<float> first, cur, prev;
    const bool verbose = getenv("REPEAT_VERBOSE") != nullptr;
    int bad_runs = 0, first_bad = -1; double max_diff = 0; long bad_words = 0;
    for (int r = 0; r < reps; r++) {
        if (whisper_decode(ctx, tok.data(), n_tok, n_past, 4) != 0) { fprintf(stderr, "decode failed\n"); return 4; }
        const float * lg = whisper_get_logits(ctx) + (size_t) (n_tok - 1) * n_vocab;      // row i belongs to token i; only the last one is fetched (src/whisper.cpp:2957-2963)
        cur.assign(lg, lg + n_vocab);
        if (r == 0) { first = cur; prev = cur; continue; }
        if (verbose) {
            long vs_prev = 0, nonfinite = 0;
            for (int i = 0; i < n_vocab; i++) { vs_prev += memcmp(&cur[i], &prev[i], 4) != 0; nonfinite += !std::isfinite(cur[i]); }
            fprintf(stderr, "rep %d: words differing from the previous rep %ld, non-finite %ld, logit[0..2] = %g %g %g (rep 0: %g %g %g)\n", r, vs_prev, nonfinite, cur[0], cur[1], cur[2], first[0], first[1], first[2]);
            prev = cur;
        }
        long bw = 0;
        for (int i = 0; i < n_vocab; i++) if (memcmp(&cur[i], &first[i], 4) != 0) { bw++; const double d = fabs((double) cur[i] - first[i]); if (d > max_diff || d != d) max_diff = d == d ? d : 1e30; }
        if (bw) { bad_runs++; bad_words += bw; if (first_bad < 0) first_bad = r; }
    }
    printf("{\"n_tokens\": %d, \"n_past\": %d, \"flash_attn\": %d, \"reps\": %d, \"mismatching_runs\": %d, \"first_bad_rep\": %d, \"mismatching_words\": %ld, \"max_abs_diff\": %.6g}\n",
           n_tok, n_past, fa, reps, bad_runs, first_bad, bad_words, max_diff);
    whisper_free(ctx);
    return bad_runs ? 1 : 0;
}
