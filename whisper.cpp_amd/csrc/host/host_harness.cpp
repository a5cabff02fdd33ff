// Native host harness above whisper.h (include/mi355x_host.h): contexts per device, states per stream, one thread per state,
// payload-skipping loader + verified weight broadcast for the replicas.  Only the reference's public API is used
// (include/whisper.h, ggml-backend.h); nothing of the reference is modified.
#include "whisper.h"
#include "ggml-backend.h"

#include "mi355x_host.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

void log_quiet(enum ggml_log_level level, const char * text, void *) { if (level == GGML_LOG_LEVEL_ERROR) fputs(text, stderr); }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- whisper_model_loader over a file that ends where the tensors begin --------------------------------------------
// legacy ggml file (src/whisper.cpp:1485-1700): u32 magic | 11 x i32 hparams | i32 n_mel, i32 n_fft, f32[n_mel*n_fft] |
// i32 n_vocab, n_vocab x (u32 len, bytes) | tensors ...
struct file_loader {
    FILE * f = nullptr;
    int64_t limit = -1;          // bytes served before end-of-file is reported (-1: the whole file)
    int64_t pos = 0, served = 0;
};
size_t fl_read(void * ctx, void * out, size_t n) {
    file_loader * l = (file_loader *) ctx;
    if (l->limit >= 0 && l->pos + (int64_t) n > l->limit) n = (size_t) std::max<int64_t>(0, l->limit - l->pos);
    const size_t r = n ? fread(out, 1, n, l->f) : 0;
    l->pos += (int64_t) r; l->served += (int64_t) r;
    return r;
}
bool fl_eof(void * ctx) { file_loader * l = (file_loader *) ctx; return (l->limit >= 0 && l->pos >= l->limit) || feof(l->f); }
void fl_close(void * ctx) { file_loader * l = (file_loader *) ctx; if (l->f) { fclose(l->f); l->f = nullptr; } }

// byte offset of the first tensor record, -1 on a malformed file
int64_t tensors_offset(const char * path, int64_t * file_size) {
    FILE * f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END); *file_size = ftell(f); fseek(f, 0, SEEK_SET);
    uint32_t magic = 0; int32_t hp[11];
    int64_t off = -1;
    do {
        if (fread(&magic, 4, 1, f) != 1 || magic != 0x67676d6c) break;
        if (fread(hp, 4, 11, f) != 11) break;
        int32_t n_mel = 0, n_fft = 0;
        if (fread(&n_mel, 4, 1, f) != 1 || fread(&n_fft, 4, 1, f) != 1 || n_mel <= 0 || n_fft <= 0 || n_mel > 4096 || n_fft > 4096) break;
        if (fseek(f, (long) n_mel * n_fft * 4, SEEK_CUR) != 0) break;
        int32_t n_vocab = 0;
        if (fread(&n_vocab, 4, 1, f) != 1 || n_vocab < 0 || n_vocab > (1 << 20)) break;
        bool ok = true;
        for (int i = 0; i < n_vocab && ok; i++) { uint32_t len = 0; ok = fread(&len, 4, 1, f) == 1 && len < (1u << 16) && fseek(f, (long) len, SEEK_CUR) == 0; }
        if (!ok) break;
        off = ftell(f);
    } while (false);
    fclose(f);
    return off;
}

whisper_context * open_model(const char * path, whisper_context_params cp, bool skip_payloads, int64_t * bytes_read, std::string & err) {
    int64_t fsz = 0;
    file_loader * fl = new file_loader();
    fl->f = fopen(path, "rb");
    if (!fl->f) { err = std::string("cannot open ") + path; delete fl; return nullptr; }
    if (skip_payloads) {
        fl->limit = tensors_offset(path, &fsz);
        if (fl->limit < 0) { err = "cannot locate the tensor section of the model file"; fclose(fl->f); delete fl; return nullptr; }
    }
    whisper_model_loader loader = { fl, fl_read, fl_eof, fl_close };
    whisper_context * ctx = whisper_init_with_params_no_state(&loader, cp);       // closes the loader itself
    if (bytes_read) *bytes_read = fl->served;
    delete fl;
    if (!ctx) err = "whisper_init_with_params_no_state failed";
    return ctx;
}

// ---- start-together barrier ---------------------------------------------------------------------------------------
struct gate {
    std::mutex m; std::condition_variable cv; int waiting = 0, generation = 0, n = 0;
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(m);
        const int g = generation;
        if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

std::vector<float> g_last_logits;
std::mutex g_last_mtx;

typedef int  (*bcast_peer_fn)(int, int, double *);
typedef int  (*bcast_rccl_group_fn)(const int *, int, double *);
typedef void (*defer_fn)(int);
typedef int  (*clone_fn)(int, int, double *);
typedef void (*set_batching_fn)(int);
typedef int  (*argmax_last_fn)(int, float *, float *);
typedef void (*batch_stats_fn)(int, uint64_t *);

} // namespace

extern "C" int mi355x_host_probe_skipping_loader(const char * model_path, int64_t * out3) {
    whisper_log_set(log_quiet, nullptr);
    int64_t fsz = 0;
    const int64_t off = tensors_offset(model_path, &fsz);
    if (off < 0) return 1;
    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = false;
    std::string err; int64_t rd = 0;
    whisper_context * ctx = open_model(model_path, cp, true, &rd, err);
    if (!ctx) return 2;
    // the reference accepts a file without tensors ("assuming empty model for testing", src/whisper.cpp:1944-1946): every tensor is
    // allocated, none is loaded
    const int ok = whisper_model_n_vocab(ctx) > 0 && whisper_model_n_text_layer(ctx) > 0;
    whisper_free(ctx);
    out3[0] = rd; out3[1] = fsz; out3[2] = off;
    return ok ? 0 : 3;
}

// One context WITH its default state (whisper_init_with_params) for hosts that drive whisper.h themselves (bench.py: one process per
// device under torchrun): skip_payloads = 1 opens it through the payload-skipping loader — the caller then fills the weights with the
// plugin's broadcast (ggml_backend_mi355x_broadcast_weights_rccl) before the first whisper_encode.
extern "C" void * mi355x_host_open(const char * model_path, int use_gpu, int gpu_device, int flash_attn, int skip_payloads, int64_t * bytes_read) {
    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = use_gpu != 0; cp.gpu_device = gpu_device; cp.flash_attn = flash_attn != 0;
    file_loader * fl = new file_loader();
    fl->f = fopen(model_path, "rb");
    if (!fl->f) { delete fl; return nullptr; }
    if (skip_payloads) {
        int64_t fsz = 0;
        fl->limit = tensors_offset(model_path, &fsz);
        if (fl->limit < 0) { fclose(fl->f); delete fl; return nullptr; }
    }
    ggml_backend_reg_t reg = use_gpu ? ggml_backend_reg_by_name("MI355X") : nullptr;
    defer_fn defer = reg ? (defer_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_defer_weights") : nullptr;
    if (skip_payloads && defer) defer(1);
    whisper_model_loader loader = { fl, fl_read, fl_eof, fl_close };
    whisper_context * ctx = whisper_init_with_params(&loader, cp);
    if (skip_payloads && defer) defer(0);
    if (bytes_read) *bytes_read = fl->served;
    delete fl;
    return ctx;
}

// one step of the bench protocol on an open context, in native code: 1 x whisper_encode + n_decode x whisper_decode(1 token) — the loop of
// examples/bench/bench.cpp:124-136 without a host-language call per token
extern "C" int mi355x_host_chunk(void * ctx_, int n_decode, int n_threads) {
    whisper_context * ctx = (whisper_context *) ctx_;
    if (!ctx) return -1;
    if (whisper_encode(ctx, 0, n_threads) != 0) return 1;
    whisper_token tok[8] = { 0 };
    for (int i = 0; i < n_decode; i++) if (whisper_decode(ctx, tok, 1, i, n_threads) != 0) return 2;
    return 0;
}

extern "C" int mi355x_host_last_logits(float * dst, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_last_mtx);
    const int64_t n = std::min<int64_t>(cap, (int64_t) g_last_logits.size());
    memcpy(dst, g_last_logits.data(), (size_t) n * 4);
    return (int) (g_last_logits.size());
}

extern "C" int mi355x_host_run(const mi355x_host_config * cfg, mi355x_host_result * out) {
    memset(out, 0, sizeof(*out));
    auto fail = [&](int code, const std::string & msg) { snprintf(out->error, sizeof(out->error), "%s", msg.c_str()); return code; };
    if (!cfg->model_path || cfg->n_devices < 1 || cfg->streams_per_device < 1 || cfg->steps < 1 || cfg->n_decode < 0) return fail(1, "bad configuration");
    whisper_log_set(log_quiet, nullptr);
    ggml_backend_reg_t reg = nullptr;
    if (cfg->use_gpu) {
        reg = ggml_backend_reg_by_name("MI355X");                  // already registered by the host application?  (loading it again would list its devices twice)
        if (!reg && cfg->plugin_path) reg = ggml_backend_load(cfg->plugin_path);
        if (!reg) return fail(2, "MI355X plugin not loaded (no CPU fallback in GPU mode)");
        if ((int) ggml_backend_reg_dev_count(reg) < cfg->first_device + (cfg->replicas_on_one_device ? 1 : cfg->n_devices)) return fail(2, "fewer MI355X devices than requested");
    }
    set_batching_fn set_batching = reg ? (set_batching_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_set_batching") : nullptr;
    batch_stats_fn  batch_stats  = reg ? (batch_stats_fn)  ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_batch_stats")  : nullptr;
    // the switch is process-wide in the plugin: whatever this run sets is put back when it ends (-1: left as it is)
    typedef int (*get_batching_fn)(void);
    get_batching_fn get_batching = reg ? (get_batching_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_get_batching") : nullptr;
    struct batching_restore { set_batching_fn set = nullptr; int old = 0; bool armed = false; ~batching_restore() { if (armed && set) set(old); } } restore;
    if (cfg->batching >= 0 && cfg->use_gpu) {
        if (!set_batching) return fail(2, "plugin has no ggml_backend_mi355x_set_batching");
        if (get_batching) { restore.set = set_batching; restore.old = get_batching(); restore.armed = true; }
        set_batching(cfg->batching);
    }
    argmax_last_fn argmax_last = reg ? (argmax_last_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_argmax_last") : nullptr;
    if (cfg->device_greedy && !(cfg->use_gpu && argmax_last)) return fail(2, "device_greedy needs the MI355X plugin's ggml_backend_mi355x_argmax_last");
    std::atomic<int64_t> g_checked{0}, g_mismatch{0};
    uint64_t bs0[5] = { 0, 0, 0, 0, 0 };
    if (batch_stats) batch_stats(cfg->first_device, bs0);
    const int nd = cfg->n_devices, ns = cfg->streams_per_device;
    const bool one_dev = cfg->use_gpu && cfg->replicas_on_one_device;
    auto device_of = [&](int r) { return cfg->first_device + (one_dev ? 0 : r); };
    out->n_devices = nd; out->streams_per_device = ns;
    int64_t fsz = 0; (void) tensors_offset(cfg->model_path, &fsz); out->file_bytes = fsz;

    // ---- contexts (one per device) and states (one per stream) ----
    const double tl0 = now_s();
    std::vector<whisper_context *> ctxs(nd, nullptr);
    std::vector<std::vector<whisper_state *>> states(nd);
    auto cleanup = [&] { for (int r = 0; r < nd; r++) { for (auto * s : states[r]) if (s) whisper_free_state(s); if (ctxs[r]) whisper_free(ctxs[r]); } };
    const bool skip = cfg->use_gpu && cfg->skip_payloads && nd > 1;
    defer_fn defer = reg ? (defer_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_defer_weights") : nullptr;
    for (int r = 0; r < nd; r++) {
        whisper_context_params cp = whisper_context_default_params();
        cp.use_gpu = cfg->use_gpu != 0; cp.gpu_device = device_of(r); cp.flash_attn = cfg->flash_attn != 0;
        std::string err; int64_t rd = 0;
        const bool sk = skip && r > 0;
        if (sk && defer) defer(1);
        ctxs[r] = open_model(cfg->model_path, cp, sk, &rd, err);
        if (sk && defer) defer(0);
        out->payload_bytes_read += rd;
        if (!ctxs[r]) { cleanup(); return fail(3, err); }
    }
    if (skip && one_dev) {
        clone_fn cl = (clone_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_clone_weights");
        if (!cl) { cleanup(); return fail(4, "plugin has no ggml_backend_mi355x_clone_weights"); }
        double st[4] = { 0, 0, 0, 0 };
        const int rc = cl(cfg->first_device, nd, st);
        out->bcast_bytes = st[0]; out->bcast_seconds = st[1]; out->bcast_buffers = (int) st[2]; out->bcast_verified = rc == 0 && st[3] == 1;
        out->bcast_transport = 3; out->bcast_ranks = 1;
        if (!out->bcast_verified) { cleanup(); return fail(4, "weight copy between the replicas of one device failed or could not be verified (rc " + std::to_string(rc) + ")"); }
    } else if (cfg->use_gpu && cfg->skip_payloads && !one_dev && ((cfg->transport == 0 && nd > 1) || cfg->transport == 2)) {
        // RCCL; transport = 2 also with ONE context (a world of one): the communicator, the grouped broadcast and the verification run on whatever hardware there is
        bcast_rccl_group_fn bc = (bcast_rccl_group_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_broadcast_weights_rccl_group");
        if (!bc) { cleanup(); return fail(4, "plugin has no ggml_backend_mi355x_broadcast_weights_rccl_group"); }
        std::vector<int> devs(nd);
        for (int r = 0; r < nd; r++) devs[r] = device_of(r);
        double st[6] = { 0, 0, 0, 0, 0, 0 };
        const int rc = bc(devs.data(), nd, st);
        out->bcast_bytes = st[0]; out->bcast_seconds = st[1]; out->bcast_buffers = (int) st[2]; out->bcast_verified = rc == 0 && st[3] == 1;
        out->bcast_setup_seconds = st[4]; out->bcast_ranks = (int) st[5]; out->bcast_transport = 1;
        if (!out->bcast_verified) { cleanup(); return fail(4, "RCCL weight broadcast failed or could not be verified (rc " + std::to_string(rc) + (rc == -1 ? ": librccl not loadable or a collective failed; transport = 1 selects peer copies" : "") + ")"); }
    } else if (skip) {
        bcast_peer_fn bc = (bcast_peer_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_broadcast_weights_peer");
        if (!bc) { cleanup(); return fail(4, "plugin has no ggml_backend_mi355x_broadcast_weights_peer"); }
        out->bcast_verified = 1; out->bcast_transport = 2; out->bcast_ranks = nd;
        for (int r = 1; r < nd; r++) {
            double st[4] = { 0, 0, 0, 0 };
            const int rc = bc(cfg->first_device, cfg->first_device + r, st);
            out->bcast_bytes += st[0]; out->bcast_seconds += st[1]; out->bcast_buffers = (int) st[2];
            if (rc != 0 || st[3] != 1) { out->bcast_verified = 0; cleanup(); return fail(4, "weight broadcast to device " + std::to_string(r) + " failed or could not be verified (rc " + std::to_string(rc) + ")"); }
        }
    }
    for (int r = 0; r < nd; r++) for (int s = 0; s < ns; s++) {
        whisper_state * st = whisper_init_state(ctxs[r]);
        if (!st) { cleanup(); return fail(5, "whisper_init_state failed"); }
        states[r].push_back(st);
    }
    out->load_s = now_s() - tl0;

    // ---- one thread per stream ----
    const int n_mels = whisper_model_n_mels(ctxs[0]), n_vocab = whisper_n_vocab(ctxs[0]), n_len = 3000;
    const int total = nd * ns;
    gate g; g.n = total + 1;
    std::atomic<int> errors{0};
    std::vector<double> t_end(total, 0.0), t_enc(total, 0.0), t_dec(total, 0.0);
    std::atomic<bool> timed{false};
    std::vector<std::vector<float>> last(total);
    std::vector<std::thread> th;
    for (int r = 0; r < nd; r++) for (int s = 0; s < ns; s++) {
        const int id = r * ns + s;
        th.emplace_back([&, r, s, id] {
            std::vector<float> mel((size_t) n_mels * n_len);
            std::mt19937 rng(1000u * (unsigned) r + 42u + (unsigned) s);
            for (auto & x : mel) x = (rng() >> 8) * (2.0f / 16777216.0f) - 1.0f;
            whisper_context * ctx = ctxs[r]; whisper_state * st = states[r][s];
            if (whisper_set_mel_with_state(ctx, st, mel.data(), n_len, n_mels) != 0) errors++;
            std::vector<whisper_token> tok(8, 0);
            auto chunk = [&] {
                const double te0 = now_s();
                if (whisper_encode_with_state(ctx, st, 0, cfg->n_threads) != 0) { errors++; return; }
                const double te1 = now_s();
                struct dec_timer { double & acc; double t0; bool on; ~dec_timer() { if (on) acc += now_s() - t0; } } dt{ t_dec[id], te1, timed.load() };
                if (dt.on) t_enc[id] += te1 - te0;
                whisper_token cur = 0;
                for (int i = 0; i < cfg->n_decode; i++) {
                    if (cfg->device_greedy) tok[0] = cur;
                    if (whisper_decode_with_state(ctx, st, tok.data(), 1, i, cfg->n_threads) != 0) { errors++; return; }
                    if (cfg->device_greedy) {
                        float top1 = 0, margin = 0;
                        const int t = argmax_last(-1, &top1, &margin);
                        if (t < 0) { errors++; return; }
                        // the same row as whisper's own sampler sees it (first maximum, strict >: src/whisper.cpp:6486-6543)
                        const float * l = whisper_get_logits_from_state(st);
                        int best = 0;
                        for (int v = 1; v < n_vocab; v++) if (l[v] > l[best]) best = v;
                        g_checked++;
                        if (best != t) g_mismatch++;
                        cur = (whisper_token) t;
                    }
                }
            };
            for (int i = 0; i < cfg->warmup; i++) chunk();
            g.arrive_and_wait();                       // everybody warm: start together (the main thread has set `timed` before it arrived)
            for (int i = 0; i < cfg->steps; i++) chunk();
            t_end[id] = now_s();
            if (cfg->n_decode > 0) { const float * l = whisper_get_logits_from_state(st); if (l) last[id].assign(l, l + n_vocab); }
        });
    }
    timed.store(true);
    g.arrive_and_wait();
    const double t0 = now_s();
    for (auto & t : th) t.join();
    double t1 = t0;
    for (double e : t_end) t1 = std::max(t1, e);
    {
        std::lock_guard<std::mutex> lk(g_last_mtx);
        g_last_logits.clear();
        for (auto & v : last) g_last_logits.insert(g_last_logits.end(), v.begin(), v.end());
    }
    cleanup();
    out->greedy_checked = g_checked.load(); out->greedy_mismatches = g_mismatch.load();
    if (batch_stats) { uint64_t bs1[5]; batch_stats(cfg->first_device, bs1); for (int i = 0; i < 5; i++) out->batch_stats[i] = bs1[i] - bs0[i]; }
    if (errors.load() != 0) return fail(6, "a whisper_encode / whisper_decode call failed");
    out->wall_s = t1 - t0;
    out->chunks_per_s = (double) total * cfg->steps / out->wall_s;
    out->ms_per_chunk_per_stream = out->wall_s * 1e3 / cfg->steps;
    double se = 0, sd = 0;
    for (int i = 0; i < total; i++) { se += t_enc[i]; sd += t_dec[i]; }
    out->encode_ms = se * 1e3 / ((double) total * cfg->steps);
    out->decode_ms_per_token = cfg->n_decode > 0 ? sd * 1e3 / ((double) total * cfg->steps * cfg->n_decode) : 0.0;
    return 0;
}
