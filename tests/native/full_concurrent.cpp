// Concurrent whisper_full() streams on ONE device (BASELINE.json configs[4]'s shape: several beam-5 / greedy streams sharing a GPU and one
// copy of the weights).  The reference's arrangement is one whisper_state per host thread on a shared whisper_context
// (src/whisper.cpp:7813-7941 whisper_full_parallel; the decoding loop and its 5-column beam step: :7076-7100, :7264).  S threads call
// whisper_full_with_state on different signals of different lengths, so that encodes, prompt steps, 5-column beam steps and single-token
// steps of different states interleave in the plugin's rendezvous (mi_batching.cpp) and on its lane streams.
//
// Checked per stream, for every batching setting asked for:
//   (a) token ids == the same stream run ALONE on the plugin;
//   (b) every logits row the sampler saw (whisper's logits_filter_callback, keyed by the decoder's token history) is BIT-identical to the
//       row of the same history in the run alone — cross-talk between states cannot hide behind an audio-independent transcript;
//   (c) token ids == the reference CPU backend's (use_gpu = false) for the same signal and parameters.
// Prints one JSON object; exit code 0 = everything held.  TEST code (links the reference libraries); with FULL_CONCURRENT_NO_CHECK=1 only the
// concurrent phase runs and nothing is compared (the ThreadSanitizer build on the stub device, tests/native/Makefile: tsan).
//   usage: full_concurrent model.bin [streams=4] [mode=greedy|beam5] [batching list, e.g. 1,2,0]
//   env GGML_MI355X_PLUGIN=path
#include "whisper.h"
#include "ggml-backend.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static void log_quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

// deterministic speech-like signal, different per stream: modulated chirps + LCG noise, 16 kHz, f32 in [-1, 1]
static std::vector<float> synth_pcm(int n, int stream) {
    std::vector<float> x(n);
    uint32_t lcg = 12345u + 7919u * (uint32_t) stream;
    const double f0 = 150 + 23 * stream, f1 = 650 + 41 * stream, f2 = 2100 + 67 * stream, ph = 0.37 * stream;
    for (int i = 0; i < n; i++) {
        const double t = i / 16000.0 + ph;
        double v = 0.35 * sin(2*M_PI*(f0 + 60*sin(2*M_PI*0.7*t))*t) + 0.20 * sin(2*M_PI*(f1 + 300*sin(2*M_PI*1.3*t))*t) + 0.10 * sin(2*M_PI*(f2 + 500*sin(2*M_PI*0.4*t))*t);
        v *= 0.5 * (1 + sin(2*M_PI*3.1*t)) * (fmod(t, 2.0) < 1.6 ? 1.0 : 0.0);
        lcg = lcg * 1664525u + 1013904223u;
        v += 0.02 * (((lcg >> 8) & 0xFFFF) / 32768.0 - 1.0);
        x[i] = (float) fmax(-1.0, fmin(1.0, v));
    }
    return x;
}

static uint64_t fnv(const void * p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char * c = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}

// rows the sampler saw, keyed by the decoder's token history (beam search calls the hook once per decoder and step, from several threads)
// a history can occur more than once (the prompt of every 30 s window; duplicate beams): the key holds the multiset of its rows' hashes
struct row_log { std::mutex m; std::map<std::pair<uint64_t, int>, std::vector<uint64_t>> rows; int n_vocab = 0; size_t n = 0;
                 void sort_all() { for (auto & kv : rows) std::sort(kv.second.begin(), kv.second.end()); } };
static void capture(struct whisper_context * ctx, struct whisper_state *, const whisper_token_data * tokens, int n_tokens, float * logits, void * ud) {
    row_log * rl = (row_log *) ud;
    std::lock_guard<std::mutex> lk(rl->m);
    if (!rl->n_vocab) rl->n_vocab = whisper_n_vocab(ctx);
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < n_tokens; i++) { h ^= (uint64_t) (uint32_t) tokens[i].id; h *= 1099511628211ull; }
    rl->rows[{ h, n_tokens }].push_back(fnv(logits, (size_t) rl->n_vocab * 4));
    rl->n++;
}

struct stream_cfg { std::vector<float> pcm; int max_tokens; };
struct stream_out { std::vector<int> tokens; row_log rows; int rc = 0; };

static int run_stream(whisper_context * ctx, whisper_state * st, const stream_cfg & c, bool beam, int n_threads, stream_out & out) {
    whisper_full_params p = whisper_full_default_params(beam ? WHISPER_SAMPLING_BEAM_SEARCH : WHISPER_SAMPLING_GREEDY);
    p.n_threads = n_threads; p.print_progress = false; p.print_realtime = false; p.print_timestamps = false; p.print_special = false;
    p.no_context = true; p.no_timestamps = true; p.single_segment = true; p.suppress_blank = false; p.suppress_nst = false;
    p.temperature = 0.0f; p.temperature_inc = 0.0f;
    p.max_tokens = c.max_tokens; p.language = "en";
    p.greedy.best_of = 1; p.beam_search.beam_size = beam ? 5 : 1;
    p.logits_filter_callback = capture; p.logits_filter_callback_user_data = &out.rows;
    out.rc = whisper_full_with_state(ctx, st, p, c.pcm.data(), (int) c.pcm.size());
    if (out.rc != 0) return out.rc;
    for (int s = 0; s < whisper_full_n_segments_from_state(st); s++)
        for (int t = 0; t < whisper_full_n_tokens_from_state(st, s); t++) out.tokens.push_back(whisper_full_get_token_id_from_state(st, s, t));
    out.rows.sort_all();
    return 0;
}

struct gate {
    std::mutex m; std::condition_variable cv; int waiting = 0, generation = 0, n = 0;
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(m);
        const int g = generation;
        if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

static void print_tokens(const std::vector<int> & t) { printf("["); for (size_t i = 0; i < t.size(); i++) printf("%s%d", i ? ", " : "", t[i]); printf("]"); }

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.bin [streams=4] [greedy|beam5] [batching list 1,2,0]\n  env GGML_MI355X_PLUGIN=path\n", argv[0]); return 2; }
    const int S = argc > 2 ? atoi(argv[2]) : 4;
    const bool beam = argc > 3 && !strcmp(argv[3], "beam5");
    std::vector<int> settings;
    { std::string l = argc > 4 ? argv[4] : "1,0"; size_t p = 0; while (p < l.size()) { settings.push_back(atoi(l.c_str() + p)); p = l.find(',', p); if (p == std::string::npos) break; p++; } }
    const bool no_check = getenv("FULL_CONCURRENT_NO_CHECK") != nullptr;
    const int n_threads = getenv("FULL_CONCURRENT_THREADS") ? atoi(getenv("FULL_CONCURRENT_THREADS")) : 2;
    whisper_log_set(log_quiet, nullptr);
    const char * plugin = getenv("GGML_MI355X_PLUGIN");
    ggml_backend_reg_t reg = plugin ? ggml_backend_load(plugin) : nullptr;
    if (!reg) { fprintf(stderr, "cannot load plugin (GGML_MI355X_PLUGIN)\n"); return 3; }
    typedef void (*set_batching_t)(int);
    typedef void (*batch_stats_t)(int, uint64_t *);
    set_batching_t set_batching = (set_batching_t) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_set_batching");
    batch_stats_t  batch_stats  = (batch_stats_t)  ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_batch_stats");
    if (!set_batching || !batch_stats) { fprintf(stderr, "plugin lacks the batching entry points\n"); return 3; }

    // signals: 6.5 .. 38 s — the long ones span two 30 s windows (a second encode while the others decode); token budgets differ so that the
    // streams leave the rendezvous at different steps
    const double secs[8] = { 11.0, 34.0, 7.5, 19.0, 38.0, 6.5, 26.0, 13.0 };
    const int    toks[8] = { 40, 28, 56, 33, 24, 64, 36, 48 };
    const int scale = getenv("FULL_CONCURRENT_TOKENS_PCT") ? atoi(getenv("FULL_CONCURRENT_TOKENS_PCT")) : 100;
    std::vector<stream_cfg> cfg(S);
    for (int s = 0; s < S; s++) { cfg[s].pcm = synth_pcm((int) (16000 * secs[s % 8]) + 160 * (s / 8), s); cfg[s].max_tokens = std::max(4, toks[s % 8] * scale / 100); }

    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = true; cp.gpu_device = 0; cp.flash_attn = true;
    whisper_context * ctx = whisper_init_from_file_with_params_no_state(argv[1], cp);
    if (!ctx) { fprintf(stderr, "model load failed\n"); return 4; }

    int bad = 0;
    printf("{\"model\": \"%s\", \"streams\": %d, \"mode\": \"%s\"", argv[1], S, beam ? "beam5" : "greedy");
    // ---- every stream ALONE on the plugin ----
    std::vector<stream_out> alone(S);
    if (!no_check) {
        set_batching(0);
        for (int s = 0; s < S; s++) {
            whisper_state * st = whisper_init_state(ctx);
            if (!st || run_stream(ctx, st, cfg[s], beam, n_threads, alone[s]) != 0) { fprintf(stderr, "stream %d alone failed\n", s); return 5; }
            whisper_free_state(st);
        }
    }
    // ---- S streams at once, per batching setting ----
    printf(",\n \"concurrent\": [");
    for (size_t k = 0; k < settings.size(); k++) {
        set_batching(settings[k]);
        uint64_t b0[5] = { 0 }, b1[5] = { 0 };
        batch_stats(0, b0);
        std::vector<whisper_state *> st(S);
        for (int s = 0; s < S; s++) { st[s] = whisper_init_state(ctx); if (!st[s]) { fprintf(stderr, "whisper_init_state failed\n"); return 5; } }
        std::vector<stream_out> out(S);
        gate g; g.n = S;
        std::vector<std::thread> th;
        for (int s = 0; s < S; s++) th.emplace_back([&, s] { g.arrive_and_wait(); run_stream(ctx, st[s], cfg[s], beam, n_threads, out[s]); });
        for (auto & t : th) t.join();
        for (int s = 0; s < S; s++) whisper_free_state(st[s]);
        batch_stats(0, b1);
        int tok_bad = 0, rows_bad = 0, rows_cmp = 0, failed = 0;
        for (int s = 0; s < S; s++) {
            if (out[s].rc != 0) { failed++; continue; }
            if (no_check) continue;
            if (out[s].tokens != alone[s].tokens) tok_bad++;
            // same histories, same rows (both runs saw the same tokens, so the key sets must coincide)
            if (out[s].rows.rows.size() != alone[s].rows.rows.size() || out[s].rows.n != alone[s].rows.n) rows_bad++;
            for (auto & kv : alone[s].rows.rows) {
                auto it = out[s].rows.rows.find(kv.first);
                rows_cmp += (int) kv.second.size();
                if (it == out[s].rows.rows.end() || it->second != kv.second) rows_bad++;
            }
        }
        const uint64_t fallbacks = b1[3] - b0[3];
        if (failed || tok_bad || rows_bad || fallbacks) bad = 1;
        printf("%s\n  {\"batching\": %d, \"failed\": %d, \"streams_with_other_tokens_than_alone\": %d, \"logit_rows_compared\": %d, \"histories_with_rows_not_bit_identical\": %d, "
               "\"merged_chains\": %llu, \"columns\": %llu, \"solo_steps\": %llu, \"fallbacks\": %llu, \"timeouts\": %llu}", k ? "," : "", settings[k], failed, tok_bad, rows_cmp, rows_bad,
               (unsigned long long) (b1[0] - b0[0]), (unsigned long long) (b1[1] - b0[1]), (unsigned long long) (b1[2] - b0[2]), (unsigned long long) fallbacks, (unsigned long long) (b1[4] - b0[4]));
    }
    printf("]");
    whisper_free(ctx);
    // ---- the reference CPU backend, stream by stream ----
    if (!no_check) {
        whisper_context_params cc = whisper_context_default_params();
        cc.use_gpu = false; cc.flash_attn = true;
        whisper_context * cctx = whisper_init_from_file_with_params_no_state(argv[1], cc);
        if (!cctx) { fprintf(stderr, "CPU model load failed\n"); return 4; }
        const int cpu_threads = getenv("FULL_CONCURRENT_CPU_THREADS") ? atoi(getenv("FULL_CONCURRENT_CPU_THREADS")) : 8;
        int cpu_bad = 0;
        printf(",\n \"per_stream\": [");
        for (int s = 0; s < S; s++) {
            stream_out ref;
            whisper_state * st = whisper_init_state(cctx);
            if (!st || run_stream(cctx, st, cfg[s], beam, cpu_threads, ref) != 0) { fprintf(stderr, "stream %d on the CPU failed\n", s); return 5; }
            whisper_free_state(st);
            size_t same = 0; while (same < ref.tokens.size() && same < alone[s].tokens.size() && ref.tokens[same] == alone[s].tokens[same]) same++;
            const bool eq = ref.tokens == alone[s].tokens;
            if (!eq) cpu_bad++;
            printf("%s\n  {\"stream\": %d, \"seconds\": %.1f, \"max_tokens\": %d, \"n_cpu\": %zu, \"n_plugin\": %zu, \"identical_prefix\": %zu, \"equal\": %s, \"cpu\": ", s ? "," : "", s,
                   cfg[s].pcm.size() / 16000.0, cfg[s].max_tokens, ref.tokens.size(), alone[s].tokens.size(), same, eq ? "true" : "false");
            print_tokens(ref.tokens);
            printf(", \"plugin\": "); print_tokens(alone[s].tokens);
            printf("}");
        }
        printf("],\n \"streams_differing_from_cpu\": %d", cpu_bad);
        if (cpu_bad) bad = 1;
        whisper_free(cctx);
    }
    printf(",\n \"ok\": %s}\n", bad ? "false" : "true");
    return bad;
}
