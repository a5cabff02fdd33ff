// Op-level parity driver: every ggml op on the whisper hot path, computed by the reference CPU backend
// (oracle/_ref/libggml-cpu.so — the reference's own kernels) and by the MI355X backend plugin loaded through
// ggml's plugin loader, on the same seeded inputs.  One JSON line per case on stdout:
//   {"case": "...", "mode": "node"|"sched", "n": N, "nmse": x, "max_abs_diff": y, "max_abs_ref": z, "mismatch_nan": k}
// mode "node"  : ggml_backend_compare_graph_backend (ggml/src/ggml-backend.cpp:2228), one kernel per ggml node.
// mode "sched" : the same graph through ggml_backend_sched with weights resident in the MI355X buffer type —
//                this is the path whisper.cpp uses, so the backend's fusion planner and hipGraph replay are active.
// This file is TEST code: it links the reference libraries, the product never does.
#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <string>
#include <vector>

static ggml_backend_t g_cpu = nullptr, g_gpu = nullptr;
static std::string g_filter;

struct rng_t {
    std::mt19937 g;
    explicit rng_t(uint32_t seed) : g(seed) {}
    float uni(float lo, float hi) { return lo + (hi - lo) * (g() >> 8) * (1.0f / 16777216.0f); }
    float nrm() { float u1 = uni(1e-7f, 1.0f), u2 = uni(0, 1); return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2); }
};

// fill a tensor (any type) from f32 values produced by gen()
static void fill_tensor(ggml_tensor * t, const std::function<float(int64_t)> & gen) {
    const int64_t n = ggml_nelements(t);
    std::vector<float> f(n);
    for (int64_t i = 0; i < n; i++) f[i] = gen(i);
    if (t->type == GGML_TYPE_F32) { ggml_backend_tensor_set(t, f.data(), 0, n * 4); return; }
    if (t->type == GGML_TYPE_I32) { std::vector<int32_t> v(n); for (int64_t i = 0; i < n; i++) v[i] = (int32_t) f[i]; ggml_backend_tensor_set(t, v.data(), 0, n * 4); return; }
    std::vector<uint8_t> q(ggml_nbytes(t));
    if (t->type == GGML_TYPE_F16) ggml_fp32_to_fp16_row(f.data(), (ggml_fp16_t *) q.data(), n);
    else ggml_quantize_chunk(t->type, f.data(), q.data(), 0, n / t->ne[0], t->ne[0], nullptr);
    ggml_backend_tensor_set(t, q.data(), 0, q.size());
}

static std::vector<float> read_f32(const ggml_tensor * t) {
    const int64_t n = ggml_nelements(t);
    std::vector<float> out(n);
    std::vector<uint8_t> raw(ggml_nbytes(t));
    ggml_backend_tensor_get(t, raw.data(), 0, raw.size());
    // tensors compared here are contiguous outputs
    if (t->type == GGML_TYPE_F32) memcpy(out.data(), raw.data(), n * 4);
    else if (t->type == GGML_TYPE_F16) ggml_fp16_to_fp32_row((const ggml_fp16_t *) raw.data(), out.data(), n);
    else { fprintf(stderr, "read_f32: unsupported type\n"); exit(2); }
    return out;
}

struct stats_t { double nmse = 0, max_diff = 0, max_ref = 0; int64_t n = 0, nan_mismatch = 0; };
static stats_t compare(const std::vector<float> & ref, const std::vector<float> & got) {
    stats_t s; s.n = (int64_t) ref.size();
    double num = 0, den = 0;
    for (size_t i = 0; i < ref.size(); i++) {
        const float a = ref[i], b = got[i];
        if (std::isnan(a) || std::isnan(b) || std::isinf(a) || std::isinf(b)) { if (!(a == b) && !(std::isnan(a) && std::isnan(b))) s.nan_mismatch++; continue; }
        const double d = (double) a - b;
        num += d * d; den += (double) a * a;
        if (fabs(d) > s.max_diff) s.max_diff = fabs(d);
        if (fabs(a) > s.max_ref) s.max_ref = fabs(a);
    }
    s.nmse = den > 0 ? num / den : num;
    return s;
}
static void report(const std::string & name, const char * mode, const stats_t & s) {
    printf("{\"case\": \"%s\", \"mode\": \"%s\", \"n\": %lld, \"nmse\": %.6e, \"max_abs_diff\": %.6e, \"max_abs_ref\": %.6e, \"mismatch_nan\": %lld}\n",
           name.c_str(), mode, (long long) s.n, s.nmse, s.max_diff, s.max_ref, (long long) s.nan_mismatch);
    fflush(stdout);
}

// a case: declare leaves with their fill functions, build the graph, name the outputs
struct builder {
    ggml_context * ctx;
    std::vector<std::pair<ggml_tensor *, std::function<float(int64_t)>>> leaves;
    rng_t rng{1234};
    ggml_tensor * leaf(ggml_type type, std::vector<int64_t> ne, std::function<float(int64_t)> gen) {
        while (ne.size() < 4) ne.push_back(1);
        ggml_tensor * t = ggml_new_tensor_4d(ctx, type, ne[0], ne[1], ne[2], ne[3]);
        leaves.push_back({ t, gen });
        return t;
    }
    ggml_tensor * randn(ggml_type type, std::vector<int64_t> ne, float scale = 1.0f) {
        const uint32_t seed = rng.g();
        auto r = std::make_shared<rng_t>(seed);
        return leaf(type, ne, [r, scale](int64_t) { return r->nrm() * scale; });
    }
};
typedef std::function<std::vector<ggml_tensor *>(builder &)> build_fn;

static void run_case(const std::string & name, const build_fn & fn, bool node_mode = true, bool sched_mode = true) {
    if (!g_filter.empty() && name.find(g_filter) == std::string::npos) return;
    const size_t meta = ggml_tensor_overhead() * 512 + ggml_graph_overhead_custom(2048, false);
    // ---- reference on CPU ----
    std::vector<std::vector<float>> ref;
    {
        ggml_init_params ip = { meta, nullptr, true };
        builder b; b.ctx = ggml_init(ip);
        std::vector<ggml_tensor *> outs = fn(b);
        ggml_cgraph * gf = ggml_new_graph_custom(b.ctx, 2048, false);
        for (auto * o : outs) ggml_build_forward_expand(gf, o);
        ggml_backend_buffer_t buf = ggml_backend_alloc_ctx_tensors(b.ctx, g_cpu);
        if (!buf) { fprintf(stderr, "%s: cpu alloc failed\n", name.c_str()); exit(2); }
        for (auto & l : b.leaves) fill_tensor(l.first, l.second);
        if (node_mode) {
            // whole graph on both backends, compare the outputs (unfused on the GPU: the copied graph carries no use counts)
            struct cb_data { std::vector<stats_t> st; } cbd;
            auto cb = [](int, ggml_tensor * t1, ggml_tensor * t2, void * ud) -> bool {
                ((cb_data *) ud)->st.push_back(compare(read_f32(t1), read_f32(t2)));
                return true;
            };
            std::vector<const ggml_tensor *> tn(outs.begin(), outs.end());
            ggml_backend_compare_graph_backend(g_cpu, g_gpu, gf, cb, &cbd, tn.data(), tn.size());
            for (size_t i = 0; i < cbd.st.size(); i++) report(name + (cbd.st.size() > 1 ? "#" + std::to_string(i) : ""), "node", cbd.st[i]);
        } else {
            ggml_backend_graph_compute(g_cpu, gf);
        }
        for (auto * o : outs) ref.push_back(read_f32(o));
        if (const char * dump = getenv("OP_PARITY_DUMP")) {
            // golden-vector export: raw leaf bytes (ggml block layout) + f32 outputs + one manifest line per case
            std::string man = "{\"case\": \"" + name + "\", \"leaves\": [";
            for (size_t i = 0; i < b.leaves.size(); i++) {
                const ggml_tensor * t = b.leaves[i].first;
                std::vector<uint8_t> raw(ggml_nbytes(t));
                ggml_backend_tensor_get(t, raw.data(), 0, raw.size());
                const std::string fn = name + ".leaf" + std::to_string(i) + ".bin";
                FILE * f = fopen((std::string(dump) + "/" + fn).c_str(), "wb"); fwrite(raw.data(), 1, raw.size(), f); fclose(f);
                char buf2[256]; snprintf(buf2, sizeof(buf2), "%s{\"file\": \"%s\", \"type\": %d, \"ne\": [%lld, %lld, %lld, %lld]}", i ? ", " : "", fn.c_str(), (int) t->type,
                                         (long long) t->ne[0], (long long) t->ne[1], (long long) t->ne[2], (long long) t->ne[3]);
                man += buf2;
            }
            man += "], \"outs\": [";
            for (size_t i = 0; i < outs.size(); i++) {
                const std::string fn = name + ".out" + std::to_string(i) + ".bin";
                FILE * f = fopen((std::string(dump) + "/" + fn).c_str(), "wb"); fwrite(ref[i].data(), 4, ref[i].size(), f); fclose(f);
                char buf2[256]; snprintf(buf2, sizeof(buf2), "%s{\"file\": \"%s\", \"ne\": [%lld, %lld, %lld, %lld]}", i ? ", " : "", fn.c_str(),
                                         (long long) outs[i]->ne[0], (long long) outs[i]->ne[1], (long long) outs[i]->ne[2], (long long) outs[i]->ne[3]);
                man += buf2;
            }
            man += "]}\n";
            FILE * f = fopen((std::string(dump) + "/manifest.jsonl").c_str(), "ab"); fputs(man.c_str(), f); fclose(f);
        }
        ggml_backend_buffer_free(buf);
        ggml_free(b.ctx);
    }
    if (!sched_mode) return;
    // ---- scheduler path: leaves in the MI355X buffer type, compute buffers planned by ggml-alloc ----
    {
        ggml_init_params ip = { meta, nullptr, true };
        builder b; b.ctx = ggml_init(ip);
        // leaves must be created (and allocated) before any op tensor: build first, then allocate only the leaves
        std::vector<ggml_tensor *> outs = fn(b);
        ggml_cgraph * gf = ggml_new_graph_custom(b.ctx, 2048, false);
        for (auto * o : outs) { ggml_set_output(o); ggml_build_forward_expand(gf, o); }
        // allocate leaves one by one in a dedicated buffer
        ggml_backend_buffer_type_t buft = ggml_backend_get_default_buffer_type(g_gpu);
        size_t total = 0; const size_t al = ggml_backend_buft_get_alignment(buft);
        for (auto & l : b.leaves) total += GGML_PAD(ggml_backend_buft_get_alloc_size(buft, l.first), al);
        ggml_backend_buffer_t wbuf = ggml_backend_buft_alloc_buffer(buft, total + al);
        if (!wbuf) { fprintf(stderr, "%s: gpu alloc failed\n", name.c_str()); exit(2); }
        ggml_tallocr ta = ggml_tallocr_new(wbuf);
        for (auto & l : b.leaves) { ggml_tallocr_alloc(&ta, l.first); fill_tensor(l.first, l.second); }
        ggml_backend_buffer_set_usage(wbuf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
        ggml_backend_t backends[2] = { g_gpu, g_cpu };
        ggml_backend_sched_t sched = ggml_backend_sched_new(backends, nullptr, 2, 2048, false, true);
        // two passes: the second one exercises hipGraph replay of the same launch sequence
        for (int pass = 0; pass < 2; pass++) {
            ggml_backend_sched_reset(sched);
            if (!ggml_backend_sched_alloc_graph(sched, gf)) { fprintf(stderr, "%s: sched alloc failed\n", name.c_str()); exit(2); }
            if (ggml_backend_sched_graph_compute(sched, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "%s: sched compute failed\n", name.c_str()); exit(2); }
            for (size_t i = 0; i < outs.size(); i++)
                report(name + (outs.size() > 1 ? "#" + std::to_string(i) : "") + (pass ? "@replay" : ""), "sched", compare(ref[i], read_f32(outs[i])));
        }
        if (getenv("OP_PARITY_SPLITS")) fprintf(stderr, "%s: splits=%d\n", name.c_str(), ggml_backend_sched_get_n_splits(sched));
        // every node on the plugin: one split, computed by the first backend.  (With GGML_MI355X_STRICT=1 an unsupported node
        // already aborts inside supports_op; this is the positive statement of the same thing.)
        if (getenv("OP_PARITY_ASSERT_SPLITS") && ggml_backend_sched_get_n_splits(sched) != 1) {
            fprintf(stderr, "%s: scheduler produced %d splits (expected 1: the whole graph on the MI355X backend)\n", name.c_str(), ggml_backend_sched_get_n_splits(sched));
            exit(6);
        }
        ggml_backend_sched_free(sched);
        ggml_backend_buffer_free(wbuf);
        ggml_free(b.ctx);
    }
}

static const char * tname(ggml_type t) { return ggml_type_name(t); }

int main(int argc, char ** argv) {
    const char * plugin = getenv("GGML_MI355X_PLUGIN");
    if (argc > 1) g_filter = argv[1];
    if (!plugin) { fprintf(stderr, "set GGML_MI355X_PLUGIN=/path/to/libggml-mi355x.so\n"); return 2; }
    if (!strcmp(plugin, "cpu")) {
        // harness self-test (no GPU needed): the reference CPU backend against itself
        g_gpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    } else {
        ggml_backend_reg_t reg = ggml_backend_load(plugin);
        if (!reg || ggml_backend_reg_dev_count(reg) == 0) { fprintf(stderr, "plugin not loaded / no MI355X device\n"); return 3; }
        g_gpu = ggml_backend_dev_init(ggml_backend_reg_dev_get(reg, 0), nullptr);
    }
    g_cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    if (!g_gpu || !g_cpu) { fprintf(stderr, "backend init failed\n"); return 3; }
    ggml_backend_cpu_set_n_threads(g_cpu, 8);
    fprintf(stderr, "op_parity: gpu backend = %s\n", ggml_backend_name(g_gpu));

    const ggml_type wtypes[] = { GGML_TYPE_Q5_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_0, GGML_TYPE_Q4_K, GGML_TYPE_F16 };

    // ---------------- small cases exported as golden vectors for the CPU oracle (tests/golden/make_golden.py) -----
    if (getenv("OP_PARITY_DUMP") || g_filter.rfind("golden", 0) == 0) {
        for (ggml_type wt : wtypes) {
            run_case(std::string("golden_mul_mat_") + tname(wt), [=](builder & b) {
                return std::vector<ggml_tensor *>{ ggml_mul_mat(b.ctx, b.randn(wt, {256, 24}, 0.07f), b.randn(GGML_TYPE_F32, {256, 3})) }; }, false, false);
        }
        run_case("golden_norm", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_norm(b.ctx, b.randn(GGML_TYPE_F32, {384, 5}, 2.0f), 1e-5f) }; }, false, false);
        run_case("golden_gelu", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_gelu(b.ctx, b.randn(GGML_TYPE_F32, {1024, 3}, 4.0f)) }; }, false, false);
        run_case("golden_soft_max", [](builder & b) {
            ggml_tensor * x = b.randn(GGML_TYPE_F32, {100, 6}, 2.0f);
            ggml_tensor * m = b.leaf(GGML_TYPE_F32, {100, 6}, [](int64_t i) { return (i % 100) > 70 + (i / 100) ? -INFINITY : 0.0f; });
            return std::vector<ggml_tensor *>{ ggml_soft_max_ext(b.ctx, x, m, 0.3f, 0.0f) }; }, false, false);
        for (int mode : { 0, 2 }) {
            run_case(std::string("golden_rope_mode") + std::to_string(mode), [=](builder & b) {
                ggml_tensor * x = b.randn(GGML_TYPE_F32, {64, 3, 7});
                ggml_tensor * pos = b.leaf(GGML_TYPE_I32, {7}, [](int64_t i) { return (float) (i * 5 + 2); });
                return std::vector<ggml_tensor *>{ ggml_rope_ext(b.ctx, x, pos, nullptr, 48, mode, 4096, 10000.0f, 0.5f, 1.0f, 1.0f, 32.0f, 1.0f) }; }, false, false);
        }
        run_case("golden_im2col", [](builder & b) {
            ggml_tensor * k = b.randn(GGML_TYPE_F16, {3, 10, 4});
            ggml_tensor * x = b.randn(GGML_TYPE_F32, {50, 10});
            return std::vector<ggml_tensor *>{ ggml_cast(b.ctx, ggml_im2col(b.ctx, k, x, 2, 0, 1, 0, 1, 0, false, GGML_TYPE_F16), GGML_TYPE_F32) }; }, false, false);
        // flash attention through the reference ("use_ref") vec path: contiguous q [D,T,H] / k,v [D,n_kv,H] views as in whisper
        run_case("golden_flash_attn", [](builder & b) {
            const int D = 64, T = 3, H = 2, n_kv = 40;
            ggml_tensor * q = b.randn(GGML_TYPE_F32, {D, H, T}, 0.7f);
            ggml_tensor * k = b.randn(GGML_TYPE_F16, {D, H, n_kv}, 0.7f);
            ggml_tensor * v = b.randn(GGML_TYPE_F16, {D, H, n_kv}, 1.0f);
            ggml_tensor * mf = b.leaf(GGML_TYPE_F32, {n_kv, T}, [=](int64_t i) { return (i % n_kv) > 30 + (i / n_kv) * 3 ? -INFINITY : 0.0f; });
            ggml_tensor * o = ggml_flash_attn_ext(b.ctx, ggml_permute(b.ctx, q, 0, 2, 1, 3), ggml_permute(b.ctx, k, 0, 2, 1, 3), ggml_permute(b.ctx, v, 0, 2, 1, 3),
                                                  ggml_cast(b.ctx, mf, GGML_TYPE_F16), 0.125f, 0.0f, 0.0f);
            return std::vector<ggml_tensor *>{ o }; }, false, false);
        // the three arithmetic paths of the CPU dispatcher (ggml-cpu/ops.cpp:9077-9230; the CPU backend runs 8 threads here):
        // split-KV (T == 1, n_kv >= 512), vec with a long F16 accumulation chain (T = 5), tiled F32 (T >= 64)
        struct fa_case { const char * name; int T, H, n_kv, mask_from; };      // mask_from < 0: no mask; else key > mask_from + 3*t is -inf
        const fa_case fa_cases[] = { { "golden_fattn_split", 1, 3, 1536, -1 }, { "golden_fattn_split_masked", 1, 2, 600, 520 },
                                     { "golden_fattn_vec5", 5, 2, 1536, -1 }, { "golden_fattn_tiled", 70, 2, 200, 90 }, { "golden_fattn_tiled_nomask", 64, 2, 136, -1 } };
        for (const fa_case & fc : fa_cases) {
            run_case(fc.name, [=](builder & b) {
                const int D = 64;
                ggml_tensor * q = b.randn(GGML_TYPE_F32, {D, fc.H, fc.T}, 0.7f);
                ggml_tensor * k = b.randn(GGML_TYPE_F16, {D, fc.H, fc.n_kv}, 0.7f);
                ggml_tensor * v = b.randn(GGML_TYPE_F16, {D, fc.H, fc.n_kv}, 1.0f);
                ggml_tensor * m = nullptr;
                if (fc.mask_from >= 0) {
                    const int n_kv = fc.n_kv, mf0 = fc.mask_from;
                    ggml_tensor * mf = b.leaf(GGML_TYPE_F32, {n_kv, fc.T}, [=](int64_t i) { return (i % n_kv) > mf0 + (i / n_kv) * 3 ? -INFINITY : 0.0f; });
                    m = ggml_cast(b.ctx, mf, GGML_TYPE_F16);
                }
                ggml_tensor * o = ggml_flash_attn_ext(b.ctx, ggml_permute(b.ctx, q, 0, 2, 1, 3), ggml_permute(b.ctx, k, 0, 2, 1, 3), ggml_permute(b.ctx, v, 0, 2, 1, 3),
                                                      m, 0.125f, 0.0f, 0.0f);
                return std::vector<ggml_tensor *>{ o }; }, false, false);
        }
        if (getenv("OP_PARITY_DUMP")) { ggml_backend_free(g_gpu); ggml_backend_free(g_cpu); return 0; }
    }

    // ---------------- mul_mat: decoder shapes (T <= 8) and encoder/prompt shapes (T > 8) ----------------
    struct mm_shape { int K, N, T; };
    const mm_shape shapes[] = {
        {1280, 1280, 1}, {1280, 1280, 5}, {1280, 5120, 1}, {5120, 1280, 2}, {512, 512, 8}, {1280, 2050, 3},
        {1280, 1280, 37}, {512, 2048, 300}, {5120, 1280, 129}, {1280, 2050, 16}, {1280, 3840, 256}, {256, 256, 1500},
    };
    for (ggml_type wt : wtypes) for (const mm_shape & s : shapes) {
        if (wt == GGML_TYPE_Q4_K && s.K % 256) continue;
        char nm[128]; snprintf(nm, sizeof(nm), "mul_mat_%s_K%d_N%d_T%d", tname(wt), s.K, s.N, s.T);
        run_case(nm, [=](builder & b) {
            ggml_tensor * w = b.randn(wt, {s.K, s.N}, 1.0f / sqrtf((float) s.K));
            ggml_tensor * x = b.randn(GGML_TYPE_F32, {s.K, s.T});
            return std::vector<ggml_tensor *>{ ggml_mul_mat(b.ctx, w, x) };
        });
    }
    // F32 weights and a K that is not a multiple of 32 (f16, conv-like)
    run_case("mul_mat_f32_K320_N96_T7", [](builder & b) {
        return std::vector<ggml_tensor *>{ ggml_mul_mat(b.ctx, b.randn(GGML_TYPE_F32, {320, 96}, 0.05f), b.randn(GGML_TYPE_F32, {320, 7})) }; });
    run_case("mul_mat_f16xf16_K240_N3000_T512", [](builder & b) {      // conv1 as GEMM for n_mels = 80 (ggml.c:4537-4565)
        return std::vector<ggml_tensor *>{ ggml_mul_mat(b.ctx, b.randn(GGML_TYPE_F16, {240, 3000}), b.randn(GGML_TYPE_F16, {240, 512}, 0.06f)) }; });
    // batched / broadcast (the -nfa attention path): K.Q with f16 K
    run_case("mul_mat_f16_batched_K64_N100_T30_H6", [](builder & b) {
        return std::vector<ggml_tensor *>{ ggml_mul_mat(b.ctx, b.randn(GGML_TYPE_F16, {64, 100, 6}), b.randn(GGML_TYPE_F32, {64, 30, 6})) }; });

    // ---------------- conv1d = im2col + mul_mat (+bias +gelu), whisper conv graph W:2012-2020 -------------
    run_case("conv1d_stride1_mels80_state384", [](builder & b) {
        ggml_tensor * k = b.randn(GGML_TYPE_F16, {3, 80, 384}, 0.06f);
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {3000, 80});
        ggml_tensor * bias = b.randn(GGML_TYPE_F32, {1, 384}, 0.02f);
        ggml_tensor * c = ggml_gelu(b.ctx, ggml_add(b.ctx, ggml_conv_1d_ph(b.ctx, k, x, 1, 1), bias));
        return std::vector<ggml_tensor *>{ c }; });
    run_case("conv1d_stride2_state384", [](builder & b) {
        ggml_tensor * k = b.randn(GGML_TYPE_F16, {3, 384, 384}, 0.03f);
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {3000, 384});
        ggml_tensor * bias = b.randn(GGML_TYPE_F32, {1, 384}, 0.02f);
        ggml_tensor * c = ggml_gelu(b.ctx, ggml_add(b.ctx, ggml_conv_1d_ph(b.ctx, k, x, 2, 1), bias));
        return std::vector<ggml_tensor *>{ c }; });
    run_case("im2col_s2", [](builder & b) {
        ggml_tensor * k = b.randn(GGML_TYPE_F16, {3, 96, 8});
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {1000, 96});
        return std::vector<ggml_tensor *>{ ggml_im2col(b.ctx, k, x, 2, 0, 1, 0, 1, 0, false, GGML_TYPE_F16) }; });

    // ---------------- flash attention (whisper's tensor views, W:2148-2170, W:2597-2625, W:2684-2705) -------
    struct fa_shape { int T, n_kv, H, n_ctx; bool mask; float scale; };
    const fa_shape fas[] = {
        {1, 1536, 20, 1536, false, 0.35355339f}, {5, 1536, 20, 1536, false, 0.35355339f}, {1, 7, 8, 448, true, 1.0f}, {5, 77, 8, 448, true, 1.0f},
        {8, 448, 6, 448, true, 1.0f}, {200, 320, 4, 320, false, 0.125f}, {256, 256, 8, 448, true, 1.0f}, {33, 100, 6, 128, true, 1.0f},
        {1500, 1536, 6, 1536, false, 0.125f},
    };
    for (const fa_shape & s : fas) {
        char nm[128]; snprintf(nm, sizeof(nm), "flash_attn_T%d_kv%d_H%d_%s", s.T, s.n_kv, s.H, s.mask ? "mask" : "nomask");
        run_case(nm, [=](builder & b) {
            const int D = 64, n_state = D * s.H;
            ggml_tensor * qcur = b.randn(GGML_TYPE_F32, {n_state, s.T}, 0.6f);
            ggml_tensor * kc = b.randn(GGML_TYPE_F16, {(int64_t) n_state * s.n_ctx}, 0.6f);
            ggml_tensor * vc = b.randn(GGML_TYPE_F16, {(int64_t) n_state * s.n_ctx}, 1.0f);
            ggml_tensor * q = ggml_permute(b.ctx, ggml_reshape_3d(b.ctx, qcur, D, s.H, s.T), 0, 2, 1, 3);
            ggml_tensor * k = ggml_view_3d(b.ctx, kc, D, s.n_kv, s.H, 2 * n_state, 2 * D, 0);
            ggml_tensor * v = ggml_view_3d(b.ctx, vc, D, s.n_kv, s.H, 2 * n_state, 2 * D, 0);
            ggml_tensor * m = nullptr;
            if (s.mask) {
                const int T = s.T, n_kv = s.n_kv;
                ggml_tensor * mf = b.leaf(GGML_TYPE_F32, {n_kv, T}, [T, n_kv](int64_t i) {
                    const int64_t t = i / n_kv, kk = i % n_kv;
                    return kk > (n_kv - T) + t ? -INFINITY : 0.0f; });     // causal: token t sees keys up to its own slot
                m = ggml_cast(b.ctx, mf, GGML_TYPE_F16);
            }
            ggml_tensor * o = ggml_flash_attn_ext(b.ctx, q, k, v, m, s.scale, 0.0f, 0.0f);
            return std::vector<ggml_tensor *>{ ggml_reshape_2d(b.ctx, o, n_state, s.T) }; });
    }

    // ---------------- bandwidth ops ---------------------------------------------------------------------------
    run_case("norm_1280x150", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_norm(b.ctx, b.randn(GGML_TYPE_F32, {1280, 150}, 3.0f), 1e-5f) }; });
    run_case("norm_affine_384x1500", [](builder & b) {
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {384, 1500}, 2.0f);
        ggml_tensor * w = b.randn(GGML_TYPE_F32, {384}, 0.1f), * bi = b.randn(GGML_TYPE_F32, {384}, 0.1f);
        return std::vector<ggml_tensor *>{ ggml_add(b.ctx, ggml_mul(b.ctx, ggml_norm(b.ctx, x, 1e-5f), w), bi) }; });
    run_case("add_bias_row", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_add(b.ctx, b.randn(GGML_TYPE_F32, {1280, 77}), b.randn(GGML_TYPE_F32, {1280})) }; });
    run_case("add_same", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_add(b.ctx, b.randn(GGML_TYPE_F32, {1282, 33}), b.randn(GGML_TYPE_F32, {1282, 33})) }; });
    run_case("add_bias_col", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_add(b.ctx, b.randn(GGML_TYPE_F32, {3000, 20}), b.randn(GGML_TYPE_F32, {1, 20})) }; });
    run_case("mul_row", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_mul(b.ctx, b.randn(GGML_TYPE_F32, {512, 9}), b.randn(GGML_TYPE_F32, {512})) }; });
    run_case("pos_emb_add_transposed", [](builder & b) {     // W:2094-2095
        ggml_tensor * cur = b.randn(GGML_TYPE_F32, {1500, 384});
        ggml_tensor * pe = b.randn(GGML_TYPE_F32, {384, 1500}, 0.02f);
        return std::vector<ggml_tensor *>{ ggml_add(b.ctx, pe, ggml_cont(b.ctx, ggml_transpose(b.ctx, cur))) }; });
    run_case("scale", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_scale(b.ctx, b.randn(GGML_TYPE_F32, {1280, 5}), 0.35355339f) }; });
    run_case("gelu", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_gelu(b.ctx, b.randn(GGML_TYPE_F32, {5120, 33}, 3.0f)) }; });
    run_case("cpy_f32_f16_view", [](builder & b) {            // KV-cache store W:2597-2598
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {512, 5});
        ggml_tensor * cache = b.randn(GGML_TYPE_F16, {512 * 448}, 1.0f);
        ggml_tensor * dst = ggml_view_1d(b.ctx, cache, 512 * 5, 2 * 512 * 17);
        return std::vector<ggml_tensor *>{ ggml_cpy(b.ctx, x, dst) }; }, true, false);
    run_case("cast_f32_f16", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_cast(b.ctx, b.randn(GGML_TYPE_F32, {77, 5}), GGML_TYPE_F16) }; });
    for (ggml_type wt : { GGML_TYPE_Q5_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_0, GGML_TYPE_Q4_K, GGML_TYPE_F16, GGML_TYPE_F32 }) {
        run_case(std::string("get_rows_") + tname(wt), [=](builder & b) {
            ggml_tensor * w = b.randn(wt, {1280, 700}, 0.05f);
            ggml_tensor * idx = b.leaf(GGML_TYPE_I32, {6}, [](int64_t i) { return (float) ((i * 131 + 5) % 700); });
            return std::vector<ggml_tensor *>{ ggml_get_rows(b.ctx, w, idx) }; });
    }
    run_case("soft_max_nomask", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_soft_max_ext(b.ctx, b.randn(GGML_TYPE_F32, {1500, 40, 3}, 2.0f), nullptr, 0.125f, 0.0f) }; });
    run_case("soft_max_mask_f32", [](builder & b) {
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {100, 32, 4}, 2.0f);
        ggml_tensor * m = b.leaf(GGML_TYPE_F32, {100, 32}, [](int64_t i) { return (i % 100) > (i / 100) + 60 ? -INFINITY : 0.0f; });
        return std::vector<ggml_tensor *>{ ggml_soft_max_ext(b.ctx, x, m, 1.0f, 0.0f) }; });
    for (int mode : { 0, 2 }) {
        run_case(std::string("rope_mode") + std::to_string(mode), [=](builder & b) {
            ggml_tensor * x = b.randn(GGML_TYPE_F32, {64, 12, 40});
            ggml_tensor * pos = b.leaf(GGML_TYPE_I32, {40}, [](int64_t i) { return (float) (i * 3 + 1); });
            return std::vector<ggml_tensor *>{ ggml_rope_ext(b.ctx, x, pos, nullptr, 64, mode, 4096, 10000.0f, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f) }; });
    }
    run_case("rope_yarn_neox_partial", [](builder & b) {
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {128, 4, 17});
        ggml_tensor * pos = b.leaf(GGML_TYPE_I32, {17}, [](int64_t i) { return (float) (i * 97 + 3); });
        return std::vector<ggml_tensor *>{ ggml_rope_ext(b.ctx, x, pos, nullptr, 96, 2, 2048, 500000.0f, 0.25f, 1.0f, 1.1f, 32.0f, 1.0f) }; });
    // multi-position rope (ggml_rope_multi: MROPE 8, VISION 24, IMROPE 40; ggml/include/ggml.h:250-254) — not on whisper's graph, op-level only
    struct mr_case { const char * name; int mode, ne0, n_dims; int sect[4]; };
    const mr_case mr_cases[] = { { "rope_mrope", 8, 128, 128, { 16, 24, 24, 0 } }, { "rope_mrope_partial", 8, 128, 64, { 8, 12, 12, 0 } }, { "rope_imrope", 40, 128, 128, { 24, 20, 20, 0 } },
                                 { "rope_vision", 24, 64, 32, { 16, 16, 0, 0 } }, { "rope_vision_4sect", 24, 80, 40, { 10, 10, 10, 10 } } };
    for (const mr_case & mc : mr_cases) {
        run_case(mc.name, [=](builder & b) {
            const int n_pos = 23;
            ggml_tensor * x = b.randn(GGML_TYPE_F32, {mc.ne0, 6, n_pos});
            ggml_tensor * pos = b.leaf(GGML_TYPE_I32, {4 * n_pos}, [=](int64_t i) { const int s = (int) (i / n_pos), p = (int) (i % n_pos); return (float) (s == 0 ? p * 3 + 1 : (s == 1 ? p / 5 + 2 : (s == 2 ? p % 5 + 7 : p + 11))); });
            int sect[4] = { mc.sect[0], mc.sect[1], mc.sect[2], mc.sect[3] };
            return std::vector<ggml_tensor *>{ ggml_rope_multi(b.ctx, x, pos, nullptr, mc.n_dims, sect, mc.mode, 4096, 10000.0f, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f) }; });
    }
    // ---------------- the ops of the voice-activity-detection graph (src/whisper.cpp:4545-4680).  whisper_vad_init_context forces
    // use_gpu = false (:4700-4704), so these never reach a GPU backend through the library: they are checked here, op by op and as the
    // graph's three sub-graphs with the silero model's shapes (tests/test-vad.cpp:31,39 is the reference's own KAT of the CPU path) ----
    run_case("unary_relu", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_relu(b.ctx, b.randn(GGML_TYPE_F32, {129, 33}, 2.0f)) }; });
    run_case("unary_sigmoid", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_sigmoid(b.ctx, b.randn(GGML_TYPE_F32, {128, 7}, 4.0f)) }; });
    run_case("unary_tanh", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_tanh(b.ctx, b.randn(GGML_TYPE_F32, {128, 7}, 3.0f)) }; });
    run_case("unary_sqrt", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_sqrt(b.ctx, b.leaf(GGML_TYPE_F32, {4, 129}, [](int64_t i) { return 0.01f * (float) (i % 977) + 1e-3f; })) }; });
    run_case("pad_reflect_1d", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_pad_reflect_1d(b.ctx, b.randn(GGML_TYPE_F32, {512, 3}), 64, 64) }; });
    run_case("vad_stft_magnitude", [](builder & b) {
        // frame [512, 1] -> reflect pad 64 -> conv1d with the STFT basis [256 taps, 1, 258] stride 128 -> magnitude of the 129 bins
        ggml_tensor * frame = b.randn(GGML_TYPE_F32, {512, 1}, 0.3f);
        ggml_tensor * basis = b.randn(GGML_TYPE_F16, {256, 1, 258}, 0.06f);
        ggml_tensor * padded = ggml_pad_reflect_1d(b.ctx, frame, 64, 64);
        ggml_tensor * stft = ggml_conv_1d(b.ctx, basis, padded, 128, 0, 1);
        const int cutoff = 129;
        ggml_tensor * re = ggml_view_2d(b.ctx, stft, 4, cutoff, stft->nb[1], 0);
        ggml_tensor * im = ggml_view_2d(b.ctx, stft, 4, cutoff, stft->nb[1], cutoff * stft->nb[1]);
        return std::vector<ggml_tensor *>{ ggml_sqrt(b.ctx, ggml_add(b.ctx, ggml_mul(b.ctx, re, re), ggml_mul(b.ctx, im, im))) }; });
    run_case("vad_encoder", [](builder & b) {
        ggml_tensor * cur = b.leaf(GGML_TYPE_F32, {4, 129}, [](int64_t i) { return 0.02f * (float) ((i * 37) % 101); });
        const int chans[5] = { 129, 128, 64, 64, 128 }, strides[4] = { 1, 2, 2, 1 };
        for (int l = 0; l < 4; l++) {
            ggml_tensor * w = b.randn(GGML_TYPE_F16, {3, chans[l], chans[l + 1]}, 1.0f / sqrtf(3.0f * chans[l]));
            ggml_tensor * bias = b.randn(GGML_TYPE_F32, {chans[l + 1]}, 0.1f);
            cur = ggml_conv_1d(b.ctx, w, cur, strides[l], 1, 1);
            cur = ggml_relu(b.ctx, ggml_add(b.ctx, cur, ggml_reshape_3d(b.ctx, bias, 1, chans[l + 1], 1)));
        }
        return std::vector<ggml_tensor *>{ cur }; });
    run_case("vad_lstm_cell", [](builder & b) {
        const int hdim = 128;
        ggml_tensor * x = b.randn(GGML_TYPE_F32, {1, 128}, 0.8f);                    // the encoder's first time step, [1, 128] view
        ggml_tensor * w_ih = b.randn(GGML_TYPE_F32, {128, 4 * hdim}, 0.09f), * b_ih = b.randn(GGML_TYPE_F32, {4 * hdim}, 0.1f);
        ggml_tensor * w_hh = b.randn(GGML_TYPE_F32, {hdim, 4 * hdim}, 0.09f), * b_hh = b.randn(GGML_TYPE_F32, {4 * hdim}, 0.1f);
        ggml_tensor * h = b.randn(GGML_TYPE_F32, {hdim}, 0.5f), * c = b.randn(GGML_TYPE_F32, {hdim}, 0.5f);
        ggml_tensor * gates = ggml_add(b.ctx, ggml_add(b.ctx, ggml_mul_mat(b.ctx, w_ih, ggml_cont(b.ctx, ggml_transpose(b.ctx, x))), b_ih),
                                       ggml_add(b.ctx, ggml_mul_mat(b.ctx, w_hh, h), b_hh));
        const size_t hs = ggml_row_size(gates->type, hdim);
        ggml_tensor * i_t = ggml_sigmoid(b.ctx, ggml_view_1d(b.ctx, gates, hdim, 0 * hs)), * f_t = ggml_sigmoid(b.ctx, ggml_view_1d(b.ctx, gates, hdim, 1 * hs));
        ggml_tensor * g_t = ggml_tanh(b.ctx, ggml_view_1d(b.ctx, gates, hdim, 2 * hs)),    * o_t = ggml_sigmoid(b.ctx, ggml_view_1d(b.ctx, gates, hdim, 3 * hs));
        ggml_tensor * c_out = ggml_add(b.ctx, ggml_mul(b.ctx, f_t, c), ggml_mul(b.ctx, i_t, g_t));
        ggml_tensor * out = ggml_mul(b.ctx, o_t, ggml_tanh(b.ctx, c_out));
        // final 1x1 convolution + sigmoid of the graph's tail
        ggml_tensor * fw = b.randn(GGML_TYPE_F16, {1, hdim, 1}, 0.2f), * fb = b.randn(GGML_TYPE_F32, {1}, 0.1f);
        ggml_tensor * prob = ggml_sigmoid(b.ctx, ggml_add(b.ctx, ggml_conv_1d(b.ctx, fw, ggml_relu(b.ctx, ggml_reshape_2d(b.ctx, out, 1, hdim)), 1, 0, 1), fb));
        return std::vector<ggml_tensor *>{ c_out, out, prob }; });
    run_case("concat_dim2", [](builder & b) { return std::vector<ggml_tensor *>{ ggml_concat(b.ctx, b.randn(GGML_TYPE_F32, {50, 7, 3}), b.randn(GGML_TYPE_F32, {50, 7, 2}), 2) }; });

    // ---------------- fused decoder / encoder sub-graphs (sched mode exercises the fusion planner) ----------------
    for (ggml_type wt : wtypes) for (int T : { 1, 5, 40 }) {
        const int n_state = 1280, n_head = 20, n_ctx = 448, kv_head = 9;
        char nm[128]; snprintf(nm, sizeof(nm), "declayer_%s_T%d", tname(wt), T);
        run_case(nm, [=](builder & b) {
            // self-attention block + MLP of whisper_build_graph_decoder (W:2529-2680, W:2770-2832) with n_kv = kv_head + T
            ggml_context * c = b.ctx;
            const int D = n_state / n_head, n_kv = kv_head + T;
            auto W = [&](int k, int n) { return b.randn(wt, {k, n}, 1.0f / sqrtf((float) k)); };
            auto V = [&](int n, float s = 0.05f) { return b.randn(GGML_TYPE_F32, {n}, s); };
            ggml_tensor * inpL = b.randn(GGML_TYPE_F32, {n_state, T}, 1.5f);
            ggml_tensor * kc = b.randn(GGML_TYPE_F16, {(int64_t) n_state * n_ctx}, 0.4f);
            ggml_tensor * vc = b.randn(GGML_TYPE_F16, {(int64_t) n_state * n_ctx}, 0.8f);
            ggml_tensor * maskf = b.leaf(GGML_TYPE_F32, {n_kv, T}, [=](int64_t i) { return (i % n_kv) > kv_head + (i / n_kv) ? -INFINITY : 0.0f; });
            ggml_tensor * mask = ggml_cast(c, maskf, GGML_TYPE_F16);
            const float KQscale = powf((float) D, -0.25f);
            ggml_tensor * cur = ggml_norm(c, inpL, 1e-5f);
            cur = ggml_add(c, ggml_mul(c, cur, V(n_state, 1.0f)), V(n_state));
            ggml_tensor * Qcur = ggml_scale(c, ggml_add(c, ggml_mul_mat(c, W(n_state, n_state), cur), V(n_state)), KQscale);
            ggml_tensor * Kcur = ggml_scale(c, ggml_mul_mat(c, W(n_state, n_state), cur), KQscale);
            ggml_tensor * Vcur = ggml_add(c, ggml_mul_mat(c, W(n_state, n_state), cur), V(n_state));
            ggml_tensor * kdst = ggml_view_1d(c, kc, (int64_t) T * n_state, (size_t) 2 * n_state * kv_head);
            ggml_tensor * vdst = ggml_view_1d(c, vc, (int64_t) T * n_state, (size_t) 2 * n_state * kv_head);
            ggml_tensor * cpk = ggml_cpy(c, Kcur, kdst), * cpv = ggml_cpy(c, Vcur, vdst);
            ggml_tensor * Q = ggml_permute(c, ggml_reshape_3d(c, Qcur, D, n_head, T), 0, 2, 1, 3);
            // make the attention depend on the cache writes (whisper orders them by graph construction; here by an explicit edge)
            ggml_tensor * Kv = ggml_view_3d(c, kc, D, n_kv, n_head, 2 * n_state, 2 * D, 0);
            ggml_tensor * Vv = ggml_view_3d(c, vc, D, n_kv, n_head, 2 * n_state, 2 * D, 0);
            Kv->src[1] = cpk; Vv->src[1] = cpv;
            ggml_tensor * att = ggml_reshape_2d(c, ggml_flash_attn_ext(c, Q, Kv, Vv, mask, 1.0f, 0.0f, 0.0f), n_state, T);
            ggml_tensor * proj = ggml_add(c, ggml_mul_mat(c, W(n_state, n_state), att), V(n_state));
            ggml_tensor * inpFF = ggml_add(c, proj, inpL);
            cur = ggml_norm(c, inpFF, 1e-5f);
            cur = ggml_add(c, ggml_mul(c, cur, V(n_state, 1.0f)), V(n_state));
            cur = ggml_gelu(c, ggml_add(c, ggml_mul_mat(c, W(n_state, 4 * n_state), cur), V(4 * n_state)));
            cur = ggml_add(c, ggml_mul_mat(c, W(4 * n_state, n_state), cur), V(n_state));
            ggml_tensor * out = ggml_add(c, cur, inpFF);
            return std::vector<ggml_tensor *>{ out, ggml_cont(c, ggml_cast(c, ggml_view_1d(c, kc, (int64_t) n_state * n_kv, 0), GGML_TYPE_F32)) };
        }, false, true);
    }

    // token + positional embedding (W:2524-2526): get_rows(quantized) + get_rows(f32), fused into one launch by the planner
    for (ggml_type wt : { GGML_TYPE_Q5_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_0, GGML_TYPE_Q4_K, GGML_TYPE_F16 }) for (int T : { 1, 5, 40 }) {
        char nm[128]; snprintf(nm, sizeof(nm), "embed_%s_T%d", tname(wt), T);
        run_case(nm, [=](builder & b) {
            ggml_tensor * te = b.randn(wt, {1280, 900}, 0.05f);
            ggml_tensor * pe = b.randn(GGML_TYPE_F32, {1280, 448}, 0.02f);
            ggml_tensor * ids = b.leaf(GGML_TYPE_I32, {T}, [](int64_t i) { return (float) ((i * 131 + 7) % 900); });
            ggml_tensor * pos = b.leaf(GGML_TYPE_I32, {T}, [](int64_t i) { return (float) (i + 3); });
            return std::vector<ggml_tensor *>{ ggml_add(b.ctx, ggml_get_rows(b.ctx, te, ids), ggml_get_rows(b.ctx, pe, pos)) };
        });
    }

    // cross-attention block of whisper_build_graph_decoder (W:2684-2770): LN -> Q (+bias, *scale) -> flash_attn over the
    // F16 cross KV (no mask) -> O-projection + bias + residual.  T = 1 takes the single-launch LN+Q+attention kernel.
    for (ggml_type wt : wtypes) for (int T : { 1, 5 }) for (int n_kv : { 1536, 200 }) {
        const int n_state = 1280, n_head = 20;
        char nm[128]; snprintf(nm, sizeof(nm), "xattnlayer_%s_T%d_kv%d", tname(wt), T, n_kv);
        run_case(nm, [=](builder & b) {
            ggml_context * c = b.ctx;
            const int D = n_state / n_head;
            auto W = [&](int k, int n) { return b.randn(wt, {k, n}, 1.0f / sqrtf((float) k)); };
            auto V = [&](int n, float s = 0.05f) { return b.randn(GGML_TYPE_F32, {n}, s); };
            ggml_tensor * inpCA = b.randn(GGML_TYPE_F32, {n_state, T}, 1.5f);
            ggml_tensor * kc = b.randn(GGML_TYPE_F16, {(int64_t) n_state * n_kv}, 0.4f);
            ggml_tensor * vc = b.randn(GGML_TYPE_F16, {(int64_t) n_state * n_kv}, 0.8f);
            const float KQscale = powf((float) D, -0.25f);
            ggml_tensor * cur = ggml_norm(c, inpCA, 1e-5f);
            cur = ggml_add(c, ggml_mul(c, cur, V(n_state, 1.0f)), V(n_state));
            ggml_tensor * Qcur = ggml_scale(c, ggml_add(c, ggml_mul_mat(c, W(n_state, n_state), cur), V(n_state)), KQscale);
            ggml_tensor * Q = ggml_permute(c, ggml_reshape_3d(c, Qcur, D, n_head, T), 0, 2, 1, 3);
            ggml_tensor * Kv = ggml_view_3d(c, kc, D, n_kv, n_head, 2 * n_state, 2 * D, 0);
            ggml_tensor * Vv = ggml_view_3d(c, vc, D, n_kv, n_head, 2 * n_state, 2 * D, 0);
            ggml_tensor * att = ggml_reshape_2d(c, ggml_flash_attn_ext(c, Q, Kv, Vv, nullptr, KQscale, 0.0f, 0.0f), n_state, T);
            ggml_tensor * proj = ggml_add(c, ggml_mul_mat(c, W(n_state, n_state), att), V(n_state));
            return std::vector<ggml_tensor *>{ ggml_add(c, proj, inpCA) };
        }, false, true);
    }

    ggml_backend_free(g_gpu);
    ggml_backend_free(g_cpu);
    return 0;
}
