// Do the decoder graphs the UNMODIFIED reference builds fit the plugin's plane pipeline and cross-state walker?  Runs on a machine
// WITHOUT a GPU: whisper decodes on the reference CPU backend, ggml_backend_sched_graph_compute is interposed (as in layer_bisect.cpp)
// and every graph is handed, before it runs, to the plugin's planner in its dry mode (ggml_backend_mi355x_debug_walk: pattern matching
// only, nothing is launched — the plugin .so is dlopen()ed directly since ggml's loader rejects it on a box without a gfx950 device).
//
//   walk_check model.bin plugin.so       -> JSON: per observed graph {n_nodes, tokens, batch verdict for S = 2 and 8, stages taken at S = 1}
// TEST code (links the reference libraries).
#include "whisper.h"
#include "ggml.h"
#include "ggml-backend.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

typedef int (*walk_fn)(void *, int, int64_t *);
static walk_fn g_walk = nullptr;
static std::string g_out;
static const char * g_tag = "";

extern "C" enum ggml_status ggml_backend_sched_graph_compute(ggml_backend_sched_t sched, struct ggml_cgraph * graph) {
    typedef enum ggml_status (*fn_t)(ggml_backend_sched_t, struct ggml_cgraph *);
    static fn_t real = (fn_t) dlsym(RTLD_NEXT, "ggml_backend_sched_graph_compute");
    if (g_walk) {
        int64_t a[6], b[6], c[6];
        g_walk(graph, 2, a); g_walk(graph, 8, b); g_walk(graph, 1, c);
        char buf[512];
        snprintf(buf, sizeof(buf), "%s  {\"what\": \"%s\", \"n_nodes\": %d, \"batch2\": %lld, \"batch8\": %lld, \"ln_stages\": %lld, \"attn_stages\": %lld, \"mm_stages\": %lld, \"other_nodes\": %lld}",
                 g_out.empty() ? "" : ",\n", g_tag, ggml_graph_n_nodes(graph), (long long) a[0], (long long) b[0], (long long) c[1], (long long) c[2], (long long) c[3], (long long) c[4]);
        g_out += buf;
    }
    return real(sched, graph);
}

static void log_quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

int main(int argc, char ** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s model.bin libggml-mi355x.so\n", argv[0]); return 2; }
    whisper_log_set(log_quiet, nullptr);
    void * h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    g_walk = (walk_fn) dlsym(h, "ggml_backend_mi355x_debug_walk");
    if (!g_walk) { fprintf(stderr, "no ggml_backend_mi355x_debug_walk in the plugin\n"); return 3; }
    whisper_context_params cp = whisper_context_default_params();
    cp.flash_attn = true; cp.use_gpu = false;
    whisper_context * ctx = whisper_init_from_file_with_params(argv[1], cp);
    if (!ctx) { fprintf(stderr, "model load failed\n"); return 3; }
    const int n_mels = whisper_model_n_mels(ctx), n_len = 3000;
    std::vector<float> mel((size_t) n_mels * n_len);
    std::mt19937 rng(7);
    for (auto & x : mel) x = (rng() >> 8) * (2.0f / 16777216.0f) - 1.0f;
    whisper_set_mel(ctx, mel.data(), n_len, n_mels);
    walk_fn keep = g_walk; g_walk = nullptr;
    if (whisper_encode(ctx, 0, 4) != 0) return 4;          // encoder graphs are not decoder steps: not asked
    g_walk = keep;
    std::vector<whisper_token> tok(8, 0);
    g_tag = "decode 1 token, n_past 0";   if (whisper_decode(ctx, tok.data(), 1, 0, 4) != 0) return 4;
    g_tag = "decode 1 token, n_past 1";   if (whisper_decode(ctx, tok.data(), 1, 1, 4) != 0) return 4;
    g_tag = "decode 1 token, n_past 200"; if (whisper_decode(ctx, tok.data(), 1, 2, 4) != 0) return 4;
    g_tag = "decode 5 tokens (beam)";     if (whisper_decode(ctx, tok.data(), 5, 0, 4) != 0) return 4;
    g_tag = "decode 8 tokens";            if (whisper_decode(ctx, tok.data(), 8, 0, 4) != 0) return 4;
    printf("[\n%s\n]\n", g_out.c_str());
    whisper_free(ctx);
    return 0;
}
