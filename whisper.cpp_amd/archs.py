"""Whisper architectures on the hot path (hyper-parameters only): the public Whisper dimensions the reference infers its
model type from (src/whisper.cpp:1525-1547); large-v3 / turbo: n_vocab 51866, n_mels 128."""

ARCHS = {
    # name: n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer, n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels
    "tiny.en":        (51864, 1500, 384, 6, 4, 448, 384, 6, 4, 80),
    "base.en":        (51864, 1500, 512, 8, 6, 448, 512, 8, 6, 80),
    "small.en":       (51864, 1500, 768, 12, 12, 448, 768, 12, 12, 80),
    "large-v3":       (51866, 1500, 1280, 20, 32, 448, 1280, 20, 32, 128),
    "large-v3-turbo": (51866, 1500, 1280, 20, 32, 448, 1280, 20, 4, 128),
    # reduced-depth variants for fast tests (same widths => same kernels / tiles)
    "large-v3-2l":    (51866, 1500, 1280, 20, 2, 448, 1280, 20, 2, 128),
    "micro":          (51864, 1500, 256, 4, 2, 448, 256, 4, 2, 80),
}

QTYPES = ("f16", "q4_0", "q5_0", "q8_0", "q4_k")
