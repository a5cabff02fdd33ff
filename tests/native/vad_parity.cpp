// The reference's only real-weight known-answer test on this path, run through the plugin.
//
// tests/test-vad.cpp:31,39 of the reference pins the voice-activity-detection graph (src/whisper.cpp:4545-4680: reflective pad, STFT as
// conv1d = im2col + F16 mul_mat, four conv layers with ReLU, an LSTM cell — F32 mul_mat on a transposed view, sigmoid / tanh gates,
// state copies —, a final 1x1 conv and a sigmoid) on TRAINED weights (models/for-tests-silero-v6.2.0-ggml.bin) and REAL speech
// (samples/jfk.wav): 344 probabilities, 4 speech segments.  The library itself never lets that graph see a GPU backend
// (whisper_vad_init_context overrides use_gpu: "GPU VAD is forced disabled", src/whisper.cpp:4700-4704), so this TEST translation unit
// includes the reference's source WHERE IT LIES ($(REF)/src/whisper.cpp, never copied — as oracle/mel_ref.cpp does) and, for the plugin
// run, re-homes what that override keeps on the CPU: the model tensors and the LSTM state go into buffers of the MI355X buffer type,
// the scheduler is rebuilt over { MI355X, CPU }, and the reference's own whisper_vad_detect_speech then runs unchanged.
//
//   vad_parity model.bin pcm_f32le.bin [cpu]      ->  one JSON object: probabilities and segments of the CPU run and of the plugin run,
//                                                     scheduler splits, nodes that did not run on the plugin
// TEST code (compiled against the reference; never shipped).
#include "src/whisper.cpp"

#include <cstdio>

static std::vector<float> read_f32(const char * path) {
    FILE * f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); const long nb = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<float> x(nb / 4);
    if (fread(x.data(), 4, x.size(), f) != x.size()) exit(2);
    fclose(f);
    return x;
}

struct vad_result { std::vector<float> probs; std::vector<std::pair<float, float>> segs; int n_splits = 0; std::vector<std::string> off_plugin; double ms = 0; };

static void run_detect(whisper_vad_context * vctx, const std::vector<float> & pcm, vad_result & r) {
    const int64_t t0 = ggml_time_us();
    if (!whisper_vad_detect_speech(vctx, pcm.data(), (int) pcm.size())) { fprintf(stderr, "whisper_vad_detect_speech failed\n"); exit(4); }
    r.ms = (ggml_time_us() - t0) * 1e-3;
    r.probs.assign(whisper_vad_probs(vctx), whisper_vad_probs(vctx) + whisper_vad_n_probs(vctx));
    whisper_vad_segments * s = whisper_vad_segments_from_probs(vctx, whisper_vad_default_params());
    for (int i = 0; i < whisper_vad_segments_n_segments(s); i++) r.segs.push_back({ whisper_vad_segments_get_segment_t0(s, i), whisper_vad_segments_get_segment_t1(s, i) });
    whisper_vad_free_segments(s);
}

// the model tensors and the LSTM state of a CPU-initialised VAD context, re-homed into MI355X buffers; scheduler over { MI355X, CPU }
static bool move_to_plugin(whisper_vad_context * vctx, std::vector<ggml_context *> & keep_ctx, std::vector<ggml_backend_buffer_t> & keep_buf) {
    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = true; cp.gpu_device = 0;
    std::vector<ggml_backend_t> backends = whisper_backend_init(cp);
    if (backends.size() < 2 || ggml_backend_dev_type(ggml_backend_get_device(backends[0])) != GGML_BACKEND_DEVICE_TYPE_GPU) {
        fprintf(stderr, "no GPU backend (is GGML_BACKEND_PATH set?)\n");
        return false;
    }
    ggml_backend_buffer_type_t buft = ggml_backend_get_default_buffer_type(backends[0]);
    whisper_vad_model & m = vctx->model;
    // weights: duplicates in a context allocated from the plugin's buffer type, contents copied, every model pointer redirected
    ggml_init_params ip = { (m.tensors.size() + 4) * ggml_tensor_overhead(), nullptr, true };
    ggml_context * wctx = ggml_init(ip);
    std::map<ggml_tensor *, ggml_tensor *> moved;
    for (auto & kv : m.tensors) { ggml_tensor * d = ggml_dup_tensor(wctx, kv.second); ggml_set_name(d, kv.first.c_str()); moved[kv.second] = d; }
    ggml_backend_buffer_t wbuf = ggml_backend_alloc_ctx_tensors_from_buft(wctx, buft);
    if (!wbuf) return false;
    ggml_backend_buffer_set_usage(wbuf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);        // as whisper does for its model buffers (src/whisper.cpp:1956)
    for (auto & kv : moved) ggml_backend_tensor_set(kv.second, kv.first->data, 0, ggml_nbytes(kv.first));
    ggml_tensor ** slots[] = { &m.stft_forward_basis, &m.encoder_0_weight, &m.encoder_0_bias, &m.encoder_1_weight, &m.encoder_1_bias, &m.encoder_2_weight, &m.encoder_2_bias,
                               &m.encoder_3_weight, &m.encoder_3_bias, &m.lstm_ih_weight, &m.lstm_ih_bias, &m.lstm_hh_weight, &m.lstm_hh_bias, &m.final_conv_weight, &m.final_conv_bias };
    for (ggml_tensor ** s : slots) { auto it = moved.find(*s); if (it == moved.end()) return false; *s = it->second; }
    for (auto & kv : m.tensors) kv.second = moved[kv.second];
    keep_ctx.push_back(wctx); keep_buf.push_back(wbuf);
    // LSTM state on the plugin
    ggml_init_params sp = { 2 * ggml_tensor_overhead(), nullptr, true };
    ggml_context * sctx = ggml_init(sp);
    vctx->h_state = ggml_new_tensor_1d(sctx, GGML_TYPE_F32, m.hparams.lstm_hidden_size); ggml_set_name(vctx->h_state, "h_state");
    vctx->c_state = ggml_new_tensor_1d(sctx, GGML_TYPE_F32, m.hparams.lstm_hidden_size); ggml_set_name(vctx->c_state, "c_state");
    ggml_backend_buffer_free(vctx->buffer);
    vctx->buffer = ggml_backend_alloc_ctx_tensors(sctx, backends[0]);
    if (!vctx->buffer) return false;
    keep_ctx.push_back(sctx);
    // scheduler over the new backend list
    ggml_backend_sched_free(vctx->sched.sched); vctx->sched.sched = nullptr;
    for (auto & b : vctx->backends) ggml_backend_free(b);
    vctx->backends = backends;
    return whisper_sched_graph_init(vctx->sched, vctx->backends, [&]() { return whisper_vad_build_graph(*vctx); });
}

int main(int argc, char ** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s model.bin pcm_f32le.bin [cpu]\n", argv[0]); return 2; }
    const bool cpu_only = argc > 3 && !strcmp(argv[3], "cpu");
    whisper_log_set([](enum ggml_log_level l, const char * t, void *) { if (l == GGML_LOG_LEVEL_ERROR) fputs(t, stderr); }, nullptr);
    ggml_backend_load_all();
    const std::vector<float> pcm = read_f32(argv[2]);
    vad_result cpu, gpu;
    {
        whisper_vad_context * v = whisper_vad_init_from_file_with_params(argv[1], whisper_vad_default_context_params());
        if (!v) return 3;
        run_detect(v, pcm, cpu);
        whisper_vad_free(v);
    }
    if (!cpu_only) {
        whisper_vad_context * v = whisper_vad_init_from_file_with_params(argv[1], whisper_vad_default_context_params());
        if (!v) return 3;
        std::vector<ggml_context *> kc; std::vector<ggml_backend_buffer_t> kb;
        if (!move_to_plugin(v, kc, kb)) { fprintf(stderr, "could not move the VAD context to the plugin\n"); return 5; }
        // where does every node of the graph run?  (one worst-case allocation, as whisper_vad_detect_speech_no_reset does)
        {
            ggml_cgraph * gf = whisper_vad_build_graph(*v);
            if (!ggml_backend_sched_alloc_graph(v->sched.sched, gf)) return 5;
            gpu.n_splits = ggml_backend_sched_get_n_splits(v->sched.sched);
            for (int i = 0; i < ggml_graph_n_nodes(gf); i++) {
                ggml_tensor * n = ggml_graph_node(gf, i);
                if (n->op == GGML_OP_NONE || n->op == GGML_OP_RESHAPE || n->op == GGML_OP_VIEW || n->op == GGML_OP_PERMUTE || n->op == GGML_OP_TRANSPOSE) continue;
                ggml_backend_t bk = ggml_backend_sched_get_tensor_backend(v->sched.sched, n);
                if (!bk || ggml_backend_dev_type(ggml_backend_get_device(bk)) != GGML_BACKEND_DEVICE_TYPE_GPU)
                    gpu.off_plugin.push_back(std::string(ggml_op_name(n->op)) + ":" + n->name);
            }
            ggml_backend_sched_reset(v->sched.sched);
        }
        run_detect(v, pcm, gpu);
        whisper_vad_free(v);                  // (frees the CPU weight buffers and the scheduler; the plugin-side copies go with the process)
    }
    auto dump = [](const char * name, const vad_result & r) {
        printf("\"%s\": {\"n_probs\": %zu, \"ms\": %.2f, \"n_splits\": %d, \"probs\": [", name, r.probs.size(), r.ms, r.n_splits);
        for (size_t i = 0; i < r.probs.size(); i++) printf("%s%.9g", i ? ", " : "", r.probs[i]);
        printf("], \"segments\": [");
        for (size_t i = 0; i < r.segs.size(); i++) printf("%s[%.2f, %.2f]", i ? ", " : "", r.segs[i].first, r.segs[i].second);
        printf("], \"off_plugin\": [");
        for (size_t i = 0; i < r.off_plugin.size(); i++) printf("%s\"%s\"", i ? ", " : "", r.off_plugin[i].c_str());
        printf("]}");
    };
    printf("{");
    dump("cpu", cpu);
    if (!cpu_only) {
        printf(", "); dump("plugin", gpu);
        double md = 0;
        for (size_t i = 0; i < cpu.probs.size() && i < gpu.probs.size(); i++) md = std::max(md, (double) fabsf(cpu.probs[i] - gpu.probs[i]));
        printf(", \"max_abs_diff\": %.6g", md);
    }
    printf("}\n");
    return 0;
}
