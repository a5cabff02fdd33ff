// TEST INFRASTRUCTURE: the reference's OWN log-mel front end, run on a PCM file.  whisper.h has no getter for the mel it computes
// (whisper_pcm_to_mel stores it inside the state), so this translation unit includes the reference's source WHERE IT LIES
// ($(REF)/src/whisper.cpp, never copied) and reads whisper_state::mel directly.  Built into oracle/_ref/mel_ref by oracle/Makefile.
//   mel_ref model.bin pcm_f32le.bin out.bin [n_threads]
//   out.bin: i32 n_mel, i32 n_len, i32 n_len_org, f32 data[n_mel * n_len]   (data[j * n_len + i], src/whisper.cpp:3046-3283)
#include "src/whisper.cpp"

#include <cstdio>

int main(int argc, char ** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model.bin pcm_f32le.bin out.bin [n_threads]\n", argv[0]); return 2; }
    const int n_threads = argc > 4 ? atoi(argv[4]) : 4;
    whisper_log_set([](enum ggml_log_level l, const char * t, void *) { if (l == GGML_LOG_LEVEL_ERROR) fputs(t, stderr); }, nullptr);
    whisper_context_params cp = whisper_context_default_params();
    cp.use_gpu = false;
    whisper_context * ctx = whisper_init_from_file_with_params(argv[1], cp);
    if (!ctx) return 3;
    FILE * f = fopen(argv[2], "rb");
    if (!f) return 4;
    fseek(f, 0, SEEK_END); const long nb = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<float> pcm(nb / 4);
    if (fread(pcm.data(), 4, pcm.size(), f) != pcm.size()) return 4;
    fclose(f);
    const int64_t t0 = ggml_time_us();
    if (whisper_pcm_to_mel(ctx, pcm.data(), (int) pcm.size(), n_threads) != 0) return 5;
    const int64_t t1 = ggml_time_us();
    const whisper_mel & mel = ctx->state->mel;
    FILE * o = fopen(argv[3], "wb");
    const int32_t hdr[3] = { mel.n_mel, mel.n_len, mel.n_len_org };
    fwrite(hdr, 4, 3, o); fwrite(mel.data.data(), 4, mel.data.size(), o); fclose(o);
    fprintf(stderr, "mel_ref: n_mel %d n_len %d n_len_org %d, %.2f ms (%d threads)\n", mel.n_mel, mel.n_len, mel.n_len_org, (t1 - t0) * 1e-3, n_threads);
    whisper_free(ctx);
    return 0;
}
