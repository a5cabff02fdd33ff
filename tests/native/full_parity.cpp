// Full-pipeline drop-in check: whisper_full() — mel front end, encoder, greedy / beam-search decoding loop, token
// timestamps bookkeeping, all of it the UNMODIFIED reference — once on the reference CPU backend and once with the MI355X
// plugin, on the same synthetic 16 kHz signal and the same model file.  Prints one JSON object with the token sequences of
// both runs.  Random-weight models produce meaningless text; what is compared is that both back ends walk the same path.
// TEST code (links the reference libraries).
#include "whisper.h"
#include "ggml-backend.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

static void log_quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

// deterministic speech-like signal: three modulated chirps + LCG noise, 16 kHz, f32 in [-1, 1]
static std::vector<float> synth_pcm(int n) {
    std::vector<float> x(n);
    uint32_t lcg = 12345;
    for (int i = 0; i < n; i++) {
        const double t = i / 16000.0;
        double v = 0.35 * sin(2*M_PI*(180 + 60*sin(2*M_PI*0.7*t))*t) + 0.20 * sin(2*M_PI*(700 + 300*sin(2*M_PI*1.3*t))*t) + 0.10 * sin(2*M_PI*(2300 + 500*sin(2*M_PI*0.4*t))*t);
        v *= 0.5 * (1 + sin(2*M_PI*3.1*t)) * (fmod(t, 2.0) < 1.6 ? 1.0 : 0.0);
        lcg = lcg * 1664525u + 1013904223u;
        v += 0.02 * (((lcg >> 8) & 0xFFFF) / 32768.0 - 1.0);
        x[i] = (float) fmax(-1.0, fmin(1.0, v));
    }
    return x;
}

// greedy runs: the logits every decoding step sampled from (whisper's own hook, include/whisper.h:474-482: called once per decoder and
// step, before the sampler's filters), so that a divergence of two back ends can be judged by the margin at the step where it happens
// beam runs: the hook is called once per DECODER and step with that decoder's token history; the history (hashed) keys the row, so that the
// rows both back ends computed for the SAME history — whichever beams each of them kept — can be compared
struct step_logits { std::vector<std::vector<float>> rows; std::vector<uint64_t> key; std::vector<int> len; int n_vocab = 0; };
static void capture_logits(struct whisper_context * ctx, struct whisper_state *, const whisper_token_data * tokens, int n_tokens, float * logits, void * ud) {
    step_logits * sl = (step_logits *) ud;
    static std::mutex mtx;                                  // (beam search processes its decoders on several threads: src/whisper.cpp:7500-7545)
    std::lock_guard<std::mutex> lk(mtx);
    if (!sl->n_vocab) sl->n_vocab = whisper_n_vocab(ctx);
    if (sl->rows.size() >= 640) return;
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < n_tokens; i++) { h ^= (uint64_t) (uint32_t) tokens[i].id; h *= 1099511628211ull; }
    sl->rows.emplace_back(logits, logits + sl->n_vocab); sl->key.push_back(h); sl->len.push_back(n_tokens);
}

static std::vector<int> run(const char * model, bool gpu, int strategy, int beam, const std::vector<float> & pcm, int max_tokens, int n_threads = 8, step_logits * cap = nullptr) {
    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = gpu; cp.gpu_device = 0; cp.flash_attn = true;
    whisper_context * ctx = whisper_init_from_file_with_params(model, cp);
    if (!ctx) { fprintf(stderr, "model load failed\n"); exit(3); }
    whisper_full_params p = whisper_full_default_params((whisper_sampling_strategy) strategy);
    p.n_threads = n_threads; p.print_progress = false; p.print_realtime = false; p.print_timestamps = false; p.print_special = false;
    p.no_context = true; p.no_timestamps = true; p.single_segment = true; p.suppress_blank = false; p.suppress_nst = false;
    p.temperature = 0.0f; p.temperature_inc = 0.0f;            // no temperature fallback: one deterministic pass
    p.max_tokens = max_tokens; p.language = "en";
    p.greedy.best_of = 1; p.beam_search.beam_size = beam;
    if (cap) { p.logits_filter_callback = capture_logits; p.logits_filter_callback_user_data = cap; }
    if (whisper_full(ctx, p, pcm.data(), (int) pcm.size()) != 0) { fprintf(stderr, "whisper_full failed\n"); exit(4); }
    std::vector<int> toks;
    for (int s = 0; s < whisper_full_n_segments(ctx); s++)
        for (int t = 0; t < whisper_full_n_tokens(ctx, s); t++) toks.push_back(whisper_full_get_token_id(ctx, s, t));
    whisper_free(ctx);
    return toks;
}

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.bin [max_tokens=24]\n  env GGML_MI355X_PLUGIN=path\n", argv[0]); return 2; }
    const int max_tokens = argc > 2 ? atoi(argv[2]) : 24;
    whisper_log_set(log_quiet, nullptr);
    const char * plugin = getenv("GGML_MI355X_PLUGIN");
    const bool selftest = plugin && !strcmp(plugin, "cpu");
    if (!selftest && (!plugin || !ggml_backend_load(plugin))) { fprintf(stderr, "cannot load plugin (GGML_MI355X_PLUGIN)\n"); return 3; }
    const std::vector<float> pcm = synth_pcm(16000 * 11);
    printf("{\"model\": \"%s\"", argv[1]);
    const struct { const char * name; int strategy, beam; } modes[] = { { "greedy", WHISPER_SAMPLING_GREEDY, 1 }, { "beam5", WHISPER_SAMPLING_BEAM_SEARCH, 5 } };
    const char * only = getenv("FULL_PARITY_ONLY");          // "greedy" / "beam5": that sampler alone (the fault-injection legs of the tests need one)
    for (const auto & m : modes) {
        if (only && strcmp(only, m.name)) continue;
        const bool greedy = m.beam == 1;
        step_logits la, lb;
        const std::vector<int> a = run(argv[1], false, m.strategy, m.beam, pcm, max_tokens, 8, &la);
        // self-test with FULL_PARITY_THREADS_B=n: reference CPU path with 8 threads against the reference CPU path with n
        // threads (different f32 summation order only) — how stable free-running decoding of this model is in the reference itself
        const int tb = selftest && getenv("FULL_PARITY_THREADS_B") ? atoi(getenv("FULL_PARITY_THREADS_B")) : 8;
        // self-test with FULL_PARITY_PERTURB=eps: the reference against itself on the signal scaled by (1 + eps)
        std::vector<float> pcm_b = pcm;
        if (selftest && getenv("FULL_PARITY_PERTURB")) { const float e = 1.0f + (float) atof(getenv("FULL_PARITY_PERTURB")); for (auto & x : pcm_b) x *= e; }
        const std::vector<int> b = run(argv[1], !selftest, m.strategy, m.beam, pcm_b, max_tokens, tb, &lb);
        size_t same = 0; while (same < a.size() && same < b.size() && a[same] == b[same]) same++;
        printf(",\n \"%s\": {\"n_cpu\": %zu, \"n_gpu\": %zu, \"identical_prefix\": %zu, ", m.name, a.size(), b.size(), same);
        if (greedy) {
            // steps 0 .. first divergence were computed on the SAME token prefix by both back ends: per step the reference's top-2 margin
            // and the largest logit difference; a divergence is explained by a near-tie when margin <= 4 x difference AT that step
            const size_t n = std::min(std::min(la.rows.size(), lb.rows.size()), same + 1);
            double min_margin = 1e30, max_diff = 0, div_margin = -1, div_diff = -1;
            for (size_t s = 0; s < n; s++) {
                const std::vector<float> & x = la.rows[s], & y = lb.rows[s];
                float t1 = -INFINITY, t2 = -INFINITY; double dmax = 0;
                for (size_t i = 0; i < x.size(); i++) {
                    if (!std::isfinite(x[i]) || !std::isfinite(y[i])) continue;
                    if (x[i] > t1) { t2 = t1; t1 = x[i]; } else if (x[i] > t2) t2 = x[i];
                    dmax = std::max(dmax, (double) fabsf(x[i] - y[i]));
                }
                min_margin = std::min(min_margin, (double) (t1 - t2)); max_diff = std::max(max_diff, dmax);
                if (s == same) { div_margin = t1 - t2; div_diff = dmax; }
            }
            printf("\"steps_compared\": %zu, \"min_margin\": %.6g, \"max_logit_diff\": %.6g, \"divergence_margin\": %.6g, \"divergence_logit_diff\": %.6g, ",
                   n, n ? min_margin : -1.0, max_diff, div_margin, div_diff);
        }
        if (!greedy) {
            // rows computed for the same decoder history on both sides (5 decoders per step; the kept beams may differ from some step on)
            size_t common = 0; int deepest = 0; double max_diff = 0;
            for (size_t i = 0; i < la.rows.size(); i++) {
                for (size_t j = 0; j < lb.rows.size(); j++) {
                    if (la.key[i] != lb.key[j] || la.len[i] != lb.len[j]) continue;
                    const std::vector<float> & x = la.rows[i], & y = lb.rows[j];
                    for (size_t k = 0; k < x.size(); k++) if (std::isfinite(x[k]) && std::isfinite(y[k])) max_diff = std::max(max_diff, (double) fabsf(x[k] - y[k]));
                    common++; deepest = std::max(deepest, la.len[i]);
                    break;
                }
            }
            printf("\"rows_cpu\": %zu, \"rows_gpu\": %zu, \"common_histories\": %zu, \"deepest_common_history\": %d, \"max_logit_diff_common\": %.6g, ",
                   la.rows.size(), lb.rows.size(), common, deepest, max_diff);
        }
        printf("\"cpu\": [");
        for (size_t i = 0; i < a.size(); i++) printf("%s%d", i ? ", " : "", a[i]);
        printf("], \"gpu\": [");
        for (size_t i = 0; i < b.size(); i++) printf("%s%d", i ? ", " : "", b[i]);
        printf("]}");
    }
    printf("}\n");
    return 0;
}
