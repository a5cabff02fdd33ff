// Model-level parity: the same whisper.cpp model file evaluated through the UNMODIFIED reference libwhisper
// once on the reference CPU backend (use_gpu = false) and once with the MI355X plugin (use_gpu = true), same
// seeded mel input, greedy decoding with teacher forcing on the CPU's tokens so that both stay on one path.
// Prints one JSON object: logits error per step, argmax agreement, and a free-running greedy transcript
// comparison.  TEST code (links the reference libraries).
#include "whisper.h"
#include "ggml-backend.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static void log_quiet(enum ggml_log_level level, const char * text, void *) {
    if (level == GGML_LOG_LEVEL_ERROR || getenv("MODEL_PARITY_VERBOSE")) fputs(text, stderr);
}

struct cmp_t { double nmse = 0, max_diff = 0, max_ref = 0; int argmax_ref = -1, argmax_got = -1; double margin_ref = 0; };

static cmp_t cmp_logits(const float * a, const float * b, int n) {
    cmp_t c; double num = 0, den = 0; float ba = -INFINITY, bb = -INFINITY, second = -INFINITY;
    for (int i = 0; i < n; i++) {
        const double d = (double) a[i] - b[i];
        num += d * d; den += (double) a[i] * a[i];
        c.max_diff = std::max(c.max_diff, fabs(d)); c.max_ref = std::max(c.max_ref, (double) fabs(a[i]));
        if (a[i] > ba) { second = ba; ba = a[i]; c.argmax_ref = i; } else if (a[i] > second) second = a[i];
        if (b[i] > bb) { bb = b[i]; c.argmax_got = i; }
    }
    c.nmse = den > 0 ? num / den : num; c.margin_ref = ba - second;
    return c;
}

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.bin [n_steps=24] [flash_attn=1]\n  env GGML_MI355X_PLUGIN=path\n", argv[0]); return 2; }
    const char * model = argv[1];
    const int n_steps = argc > 2 ? atoi(argv[2]) : 24;
    const bool fa = argc > 3 ? atoi(argv[3]) != 0 : true;
    const int n_threads = getenv("MODEL_PARITY_THREADS") ? atoi(getenv("MODEL_PARITY_THREADS")) : 8;
    whisper_log_set(log_quiet, nullptr);
    const char * plugin = getenv("GGML_MI355X_PLUGIN");
    const bool selftest = plugin && !strcmp(plugin, "cpu");     // harness self-test: CPU against CPU
    if (!selftest && (!plugin || !ggml_backend_load(plugin))) { fprintf(stderr, "cannot load plugin (GGML_MI355X_PLUGIN)\n"); return 3; }

    whisper_context_params cp = whisper_context_default_params();
    cp.flash_attn = fa;
    cp.use_gpu = false;
    whisper_context * cpu = whisper_init_from_file_with_params(model, cp);
    cp.use_gpu = !selftest; cp.gpu_device = 0;
    // self-test with MODEL_PARITY_SPREAD=1: reference CPU path with flash attention against the reference CPU path
    // without it — the reference's OWN spread between two of its arithmetic paths, to put the GPU tolerance in context
    if (selftest && getenv("MODEL_PARITY_SPREAD")) cp.flash_attn = !fa;
    whisper_context * gpu = whisper_init_from_file_with_params(model, cp);
    if (!cpu || !gpu) { fprintf(stderr, "model load failed\n"); return 3; }

    const int n_mels = whisper_model_n_mels(cpu), n_vocab = whisper_n_vocab(cpu), n_len = 3000;
    std::vector<float> mel((size_t) n_mels * n_len);
    const float mel_phase = getenv("MODEL_PARITY_MEL_SEED") ? 0.37f * (float) atoi(getenv("MODEL_PARITY_MEL_SEED")) : 0.0f;
    std::mt19937 rng(getenv("MODEL_PARITY_MEL_SEED") ? (unsigned) atoi(getenv("MODEL_PARITY_MEL_SEED")) : 42u);      // another seed = another "audio"
    for (int j = 0; j < n_mels; j++) for (int i = 0; i < n_len; i++)
        mel[(size_t) j * n_len + i] = 0.6f * sinf(0.013f * i + 0.21f * j + mel_phase) + 0.4f * ((rng() >> 8) * (1.0f / 8388608.0f) - 1.0f);
    whisper_set_mel(cpu, mel.data(), n_len, n_mels);
    // self-test with MODEL_PARITY_PERTURB=eps: the reference against ITSELF on an input scaled by (1 + eps) — how far the reference's
    // own logits move under a perturbation of the size of one f32 rounding (its int8 activation rounding decides discretely)
    if (selftest && getenv("MODEL_PARITY_PERTURB")) { const float e = 1.0f + (float) atof(getenv("MODEL_PARITY_PERTURB")); for (auto & x : mel) x *= e; }
    whisper_set_mel(gpu, mel.data(), n_len, n_mels);

    auto t0 = std::chrono::steady_clock::now();
    if (whisper_encode(cpu, 0, n_threads) != 0) { fprintf(stderr, "cpu encode failed\n"); return 4; }
    auto t1 = std::chrono::steady_clock::now();
    if (whisper_encode(gpu, 0, n_threads) != 0) { fprintf(stderr, "gpu encode failed\n"); return 4; }
    auto t2 = std::chrono::steady_clock::now();
    if (whisper_encode(gpu, 0, n_threads) != 0) { fprintf(stderr, "gpu encode failed\n"); return 4; }
    auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };

    printf("{\"model\": \"%s\", \"flash_attn\": %d, \"encode_ms_cpu\": %.2f, \"encode_ms_gpu_first\": %.2f, \"encode_ms_gpu\": %.2f,\n", model, fa ? 1 : 0, ms(t0, t1), ms(t1, t2), ms(t2, t3));

    // ---- teacher-forced single-token steps ----
    double worst_nmse = 0, worst_diff = 0, max_ref = 0, min_margin_on_mismatch = 1e30, max_margin_on_mismatch = 0, sum_nmse = 0; int agree = 0;
    int n_tight = 0, n_tight_mismatch = 0;      // steps whose CPU top-2 margin is below 2 x the largest logit difference of that step ("near-ties")
    const bool brief = n_steps > 32 && !getenv("MODEL_PARITY_ALL_STEPS");      // long runs: per-step rows only for the steps that matter
    whisper_token tok = whisper_token_sot(cpu);
    double gpu_ms = 0, cpu_ms = 0;
    printf(" \"steps\": [");
    for (int i = 0; i < n_steps; i++) {
        auto a0 = std::chrono::steady_clock::now();
        if (whisper_decode(cpu, &tok, 1, i, n_threads) != 0) { fprintf(stderr, "cpu decode failed\n"); return 4; }
        auto a1 = std::chrono::steady_clock::now();
        if (whisper_decode(gpu, &tok, 1, i, n_threads) != 0) { fprintf(stderr, "gpu decode failed\n"); return 4; }
        auto a2 = std::chrono::steady_clock::now();
        cpu_ms += ms(a0, a1); gpu_ms += ms(a1, a2);
        const cmp_t c = cmp_logits(whisper_get_logits(cpu), whisper_get_logits(gpu), n_vocab);
        worst_nmse = std::max(worst_nmse, c.nmse); worst_diff = std::max(worst_diff, c.max_diff); max_ref = std::max(max_ref, c.max_ref);
        sum_nmse += c.nmse;
        const bool tight = c.margin_ref < 2.0 * c.max_diff;
        if (tight) n_tight++;
        if (c.argmax_ref == c.argmax_got) agree++;
        else { min_margin_on_mismatch = std::min(min_margin_on_mismatch, c.margin_ref); max_margin_on_mismatch = std::max(max_margin_on_mismatch, c.margin_ref / std::max(c.max_diff, 1e-30)); if (tight) n_tight_mismatch++; }
        static bool first_row = true;
        if (!brief || tight || c.argmax_ref != c.argmax_got || i < 4) {
            printf("%s{\"step\": %d, \"nmse\": %.3e, \"max_diff\": %.3e, \"tok_cpu\": %d, \"tok_gpu\": %d, \"margin\": %.3e}", first_row ? "" : ", ", i, c.nmse, c.max_diff, c.argmax_ref, c.argmax_got, c.margin_ref);
            first_row = false;
        }
        tok = c.argmax_ref;
        if (tok >= whisper_token_eot(cpu)) tok = (whisper_token) (i * 7919 % 50000);    // keep decoding text tokens
    }
    // mismatch statistics: a teacher-forced argmax may only differ where the CPU's own top-2 margin is within the logit difference
    printf("],\n \"single\": {\"steps\": %d, \"argmax_agree\": %d, \"worst_nmse\": %.3e, \"mean_nmse\": %.3e, \"worst_abs_diff\": %.3e, \"max_abs_logit\": %.3e, \"min_margin_on_mismatch\": %.3e, "
           "\"max_margin_over_maxdiff_on_mismatch\": %.3f, \"near_tie_steps\": %d, \"near_tie_mismatches\": %d, \"ms_per_tok_cpu\": %.3f, \"ms_per_tok_gpu\": %.3f},\n",
           n_steps, agree, worst_nmse, sum_nmse / n_steps, worst_diff, max_ref, min_margin_on_mismatch > 1e29 ? -1.0 : min_margin_on_mismatch,
           max_margin_on_mismatch, n_tight, n_tight_mismatch, cpu_ms / n_steps, gpu_ms / n_steps);

    // ---- batches: 5 tokens (beam-sized) and a 48-token prompt, n_past = 0 ----
    for (int nb : { 5, 48 }) {
        std::vector<whisper_token> toks(nb);
        for (int i = 0; i < nb; i++) toks[i] = (whisper_token) ((i * 2654435761u + 17) % 50000);
        if (whisper_decode(cpu, toks.data(), nb, 0, n_threads) != 0 || whisper_decode(gpu, toks.data(), nb, 0, n_threads) != 0) { fprintf(stderr, "batch decode failed\n"); return 4; }
        double wn = 0, wd = 0; int ag = 0;
        // row i of the logits buffer belongs to token i (whisper.cpp:2957-2963): compare the last token's row
        const size_t off = (size_t) (nb - 1) * n_vocab;
        const cmp_t c = cmp_logits(whisper_get_logits(cpu) + off, whisper_get_logits(gpu) + off, n_vocab);
        wn = c.nmse; wd = c.max_diff; ag = c.argmax_ref == c.argmax_got;
        printf(" \"batch%d\": {\"nmse\": %.3e, \"max_diff\": %.3e, \"argmax_agree\": %d},\n", nb, wn, wd, ag);
    }

    // ---- free-running greedy: each side follows its own argmax (text tokens only), in lockstep.  While the histories are identical
    // the two logit rows belong to the same input, so the first divergence can be judged like a teacher-forced step: it is
    // legitimate only at a near-tie of the CPU's own top-2 candidates ----
    const int n_free = std::min(n_steps, 200);
    int same = 0; double div_margin = -1, div_maxdiff = -1;
    {
        whisper_token tc = whisper_token_sot(cpu), tg = tc;
        const int n_text = whisper_token_eot(cpu);
        for (int i = 0; i < n_free; i++) {
            if (whisper_decode(cpu, &tc, 1, i, n_threads) != 0 || whisper_decode(gpu, &tg, 1, i, n_threads) != 0) return 4;
            const cmp_t c = cmp_logits(whisper_get_logits(cpu), whisper_get_logits(gpu), n_text);
            if (c.argmax_ref != c.argmax_got) { div_margin = c.margin_ref; div_maxdiff = c.max_diff; break; }
            same++; tc = tg = c.argmax_ref;
        }
    }
    printf(" \"greedy\": {\"steps\": %d, \"identical_prefix\": %d, \"divergence_margin\": %.3e, \"divergence_max_diff\": %.3e}}\n", n_free, same, div_margin, div_maxdiff);

    whisper_free(cpu); whisper_free(gpu);
    return 0;
}
