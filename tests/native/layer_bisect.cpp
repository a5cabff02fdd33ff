// Where does the CPU-vs-plugin difference of a decode step come from?  Runs the same model file through the UNMODIFIED libwhisper
// on the reference CPU backend and on the MI355X plugin and compares EVERY node of the decoder graph (ggml's own per-node hook,
// ggml_backend_sched_set_eval_callback, ggml-backend.h:316,354), not just the logits.  whisper.h does not expose its scheduler,
// so this test binary interposes ggml_backend_sched_graph_compute (the one call whisper makes per graph, src/whisper.cpp:206)
// and installs the callback on the scheduler it is handed.
//
//   layer_bisect model.bin [n_tokens=5] [n_past=0]      env: GGML_MI355X_PLUGIN, BISECT_THREADS (8), BISECT_ALL=1 (print every node)
//
// Output: one JSON object — per op class the largest single-node error JUMP (node NMSE / largest NMSE among its inputs),
// the NMSE after every decoder layer, and the logits NMSE.  TEST code (links the reference libraries).
#include "whisper.h"
#include "ggml.h"
#include "ggml-backend.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

// tensor structs live in whisper's per-graph meta arena: everything that needs them is resolved while the graph is alive; afterwards
// the pointers are identity keys only
struct node_rec { std::string op, name; int64_t ne[4]; std::vector<float> data; const void * in_keys[4]; const void * self; bool residual; };
struct graph_rec { int n_nodes = 0; std::vector<node_rec> nodes; };

static bool g_observe = false;
static std::vector<graph_rec> * g_sink = nullptr;      // graphs of the current run
static int64_t g_max_cols = 64;

static bool is_view_op(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE;
}

static bool eval_cb(ggml_tensor * t, bool ask, void *) {
    const bool want = !is_view_op(t) && t->type == GGML_TYPE_F32 && ggml_is_contiguous(t) && ggml_nrows(t) <= g_max_cols * 64 && ggml_nelements(t) <= (int64_t) 4 << 20;
    if (ask) return want;
    if (!want) return true;
    node_rec r;
    r.op = ggml_op_desc(t); r.name = t->name; r.self = t;
    for (int i = 0; i < 4; i++) {
        r.ne[i] = t->ne[i];
        const ggml_tensor * s = t->src[i];        // follow views / copies back to the tensor that was computed
        while (s && (is_view_op(s) || s->op == GGML_OP_CPY || s->op == GGML_OP_CONT) && s->src[0]) s = s->src[0];
        r.in_keys[i] = s;
    }
    r.residual = t->op == GGML_OP_ADD && t->src[0] && t->src[1] && ggml_are_same_shape(t->src[0], t->src[1]);
    r.data.resize(ggml_nelements(t));
    ggml_backend_tensor_get(t, r.data.data(), 0, ggml_nbytes(t));
    g_sink->back().nodes.push_back(std::move(r));
    return true;
}

extern "C" enum ggml_status ggml_backend_sched_graph_compute(ggml_backend_sched_t sched, struct ggml_cgraph * graph) {
    typedef enum ggml_status (*fn_t)(ggml_backend_sched_t, struct ggml_cgraph *);
    static fn_t real = (fn_t) dlsym(RTLD_NEXT, "ggml_backend_sched_graph_compute");
    if (g_observe && g_sink) {
        g_sink->emplace_back();
        g_sink->back().n_nodes = ggml_graph_n_nodes(graph);
        ggml_backend_sched_set_eval_callback(sched, eval_cb, nullptr);
    } else {
        ggml_backend_sched_set_eval_callback(sched, nullptr, nullptr);
    }
    return real(sched, graph);
}

static void log_quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

static double nmse(const std::vector<float> & a, const std::vector<float> & b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) { const double d = (double) a[i] - b[i]; num += d * d; den += (double) a[i] * a[i]; }
    return den > 0 ? num / den : num;
}

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.bin [n_tokens=5] [n_past=0]\n", argv[0]); return 2; }
    const int n_tok = argc > 2 ? atoi(argv[2]) : 5, n_past = argc > 3 ? atoi(argv[3]) : 0;
    const int n_threads = getenv("BISECT_THREADS") ? atoi(getenv("BISECT_THREADS")) : 8;
    whisper_log_set(log_quiet, nullptr);
    const char * plugin = getenv("GGML_MI355X_PLUGIN");
    const bool selftest = plugin && !strcmp(plugin, "cpu");
    if (!selftest && (!plugin || !ggml_backend_load(plugin))) { fprintf(stderr, "cannot load plugin (GGML_MI355X_PLUGIN)\n"); return 3; }
    g_max_cols = std::max(64, n_tok);

    std::vector<graph_rec> runs[2];
    std::vector<float> logits[2];
    int n_vocab = 0;
    for (int side = 0; side < 2; side++) {
        whisper_context_params cp = whisper_context_default_params();
        cp.flash_attn = true; cp.use_gpu = side == 1 && !selftest; cp.gpu_device = 0;
        whisper_context * ctx = whisper_init_from_file_with_params(argv[1], cp);
        if (!ctx) { fprintf(stderr, "model load failed\n"); return 3; }
        const int n_mels = whisper_model_n_mels(ctx), n_len = 3000;
        n_vocab = whisper_n_vocab(ctx);
        std::vector<float> mel((size_t) n_mels * n_len);
        std::mt19937 rng(42);
        for (int j = 0; j < n_mels; j++) for (int i = 0; i < n_len; i++)
            mel[(size_t) j * n_len + i] = 0.6f * sinf(0.013f * i + 0.21f * j) + 0.4f * ((rng() >> 8) * (1.0f / 8388608.0f) - 1.0f);
        whisper_set_mel(ctx, mel.data(), n_len, n_mels);
        // self-test with BISECT_THREADS_B: the reference against itself at another thread count (its split-KV attention depends on it)
        const int nt = (side == 1 && selftest && getenv("BISECT_THREADS_B")) ? atoi(getenv("BISECT_THREADS_B")) : n_threads;
        if (whisper_encode(ctx, 0, nt) != 0) { fprintf(stderr, "encode failed\n"); return 4; }
        std::vector<whisper_token> toks(n_tok);
        for (int i = 0; i < n_tok; i++) toks[i] = (whisper_token) ((i * 2654435761u + 17) % 50000);
        if (n_past > 0) {           // fill the self-attention cache first (unobserved)
            std::vector<whisper_token> pre(n_past);
            for (int i = 0; i < n_past; i++) pre[i] = (whisper_token) ((i * 40503u + 5) % 50000);
            if (whisper_decode(ctx, pre.data(), n_past, 0, nt) != 0) { fprintf(stderr, "prefill failed\n"); return 4; }
        }
        g_sink = &runs[side]; g_observe = true;
        if (whisper_decode(ctx, toks.data(), n_tok, n_past, nt) != 0) { fprintf(stderr, "decode failed\n"); return 4; }
        g_observe = false; g_sink = nullptr;
        const float * l = whisper_get_logits(ctx) + (size_t) (n_tok - 1) * n_vocab;
        logits[side].assign(l, l + n_vocab);
        whisper_free(ctx);
    }
    if (runs[0].size() != runs[1].size()) { fprintf(stderr, "graph count differs: %zu vs %zu\n", runs[0].size(), runs[1].size()); return 5; }

    printf("{\"model\": \"%s\", \"n_tokens\": %d, \"n_past\": %d, \"threads\": %d, \"exact_mode\": %d,\n", argv[1], n_tok, n_past, n_threads, getenv("GGML_MI355X_EXACT") ? atoi(getenv("GGML_MI355X_EXACT")) : 0);
    struct cls { double worst_jump = 0, worst_nmse = 0, sum_nmse = 0; int n = 0; std::string where; };
    std::map<std::string, cls> by_op;
    std::vector<std::pair<int, double>> residual_curve;
    const bool all = getenv("BISECT_ALL") != nullptr;
    for (size_t gi = 0; gi < runs[0].size(); gi++) {
        const graph_rec & A = runs[0][gi], & B = runs[1][gi];
        if (A.nodes.size() != B.nodes.size()) { fprintf(stderr, "graph %zu: observed node count differs (%zu vs %zu)\n", gi, A.nodes.size(), B.nodes.size()); return 5; }
        std::map<const void *, double> err;                  // CPU-side tensor -> NMSE of the plugin's value
        int n_res = 0;
        for (size_t i = 0; i < A.nodes.size(); i++) {
            const node_rec & a = A.nodes[i], & b = B.nodes[i];
            if (a.op != b.op || a.data.size() != b.data.size()) { fprintf(stderr, "graph %zu node %zu: %s vs %s\n", gi, i, a.op.c_str(), b.op.c_str()); return 5; }
            const double e = nmse(a.data, b.data);
            err[a.self] = e;
            // error of the inputs: follow view chains back to observed producers
            double ein = 0;
            for (int s = 0; s < 4; s++) if (a.in_keys[s] && err.count(a.in_keys[s])) ein = std::max(ein, err[a.in_keys[s]]);
            cls & c = by_op[a.op];
            c.n++; c.sum_nmse += e; c.worst_nmse = std::max(c.worst_nmse, e);
            const double jump = e / std::max(ein, 1e-14);
            if (ein > 0 && jump > c.worst_jump) { c.worst_jump = jump; char buf[160]; snprintf(buf, sizeof(buf), "graph %zu node %zu [%lld,%lld,%lld]: %.2e -> %.2e", gi, i, (long long) a.ne[0], (long long) a.ne[1], (long long) a.ne[2], ein, e); c.where = buf; }
            if (all) fprintf(stderr, "g%zu n%zu %-16s [%lld,%lld,%lld,%lld] in %.3e out %.3e %s\n", gi, i, a.op.c_str(), (long long) a.ne[0], (long long) a.ne[1], (long long) a.ne[2], (long long) a.ne[3], ein, e, a.name.c_str());
            // residual stream: ADD of two full [n_state, T] tensors
            if (a.residual && a.ne[1] == n_tok && gi + 1 == runs[0].size()) residual_curve.push_back({ n_res++, e });
        }
    }
    printf(" \"per_op\": {");
    bool first = true;
    for (auto & kv : by_op) {
        printf("%s\n  \"%s\": {\"nodes\": %d, \"mean_nmse\": %.3e, \"worst_nmse\": %.3e, \"worst_jump_x\": %.1f, \"at\": \"%s\"}", first ? "" : ",", kv.first.c_str(), kv.second.n,
               kv.second.sum_nmse / std::max(kv.second.n, 1), kv.second.worst_nmse, kv.second.worst_jump, kv.second.where.c_str());
        first = false;
    }
    printf("},\n \"residual_stream_nmse\": [");
    for (size_t i = 0; i < residual_curve.size(); i++) printf("%s%.3e", i ? ", " : "", residual_curve[i].second);
    printf("],\n \"logits_nmse\": %.3e}\n", nmse(logits[0], logits[1]));
    return 0;
}
