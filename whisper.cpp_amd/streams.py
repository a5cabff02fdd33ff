"""Several concurrent 30 s streams on ONE GPU: one whisper_state per stream on a shared
whisper_context, i.e. one set of weights in HBM, one ggml backend instance (own HIP stream, own KV caches and compute
buffers) per stream, one host thread per stream — the arrangement whisper_full_parallel uses (src/whisper.cpp:7848-7869).
A decode step is a chain of ~230 dependent, latency-bound launches, so the GPU is mostly idle inside one stream; streams on
different HIP streams fill those holes.  Everything here is the unmodified whisper.h API through ctypes (which drops the
GIL during every call).
"""
from __future__ import annotations

import ctypes as C
import sys
import threading


def bind(w):
    """declare the *_with_state entry points of include/whisper.h on a loaded libwhisper"""
    w.whisper_init_state.restype = C.c_void_p
    w.whisper_init_state.argtypes = [C.c_void_p]
    w.whisper_free_state.argtypes = [C.c_void_p]
    w.whisper_set_mel_with_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    w.whisper_encode_with_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    w.whisper_decode_with_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    w.whisper_get_logits_from_state.restype = C.POINTER(C.c_float)
    w.whisper_get_logits_from_state.argtypes = [C.c_void_p]
    w.whisper_n_vocab.argtypes = [C.c_void_p]
    return w


class Streams:
    def __init__(self, w, ctx, n_streams: int, mels, n_threads: int = 2):
        """mels: one float32 [n_mels, 3000] array per stream"""
        self.w, self.ctx, self.n, self.n_threads = bind(w), ctx, n_streams, n_threads
        self.states = []
        for s in range(n_streams):
            st = w.whisper_init_state(ctx)
            if not st:
                raise RuntimeError("whisper_init_state failed")
            self.states.append(st)
            m = mels[s]
            if w.whisper_set_mel_with_state(ctx, st, m.ctypes.data_as(C.c_void_p), m.shape[1], m.shape[0]) != 0:
                raise RuntimeError("whisper_set_mel_with_state failed")
        self.tokens = (C.c_int32 * 512)()
        sys.setswitchinterval(1e-5)

    def chunk_one(self, s: int, n_decode: int, tokens=None, keep_logits=None):
        """1 x encode + n_decode x single-token decode on stream s (the whisper-bench protocol)"""
        w, ctx, st = self.w, self.ctx, self.states[s]
        toks = tokens if tokens is not None else self.tokens
        if w.whisper_encode_with_state(ctx, st, 0, self.n_threads) != 0:
            raise RuntimeError("whisper_encode_with_state failed")
        for i in range(n_decode):
            if w.whisper_decode_with_state(ctx, st, C.byref(toks, 4 * i) if tokens is not None else toks, 1, i, self.n_threads) != 0:
                raise RuntimeError("whisper_decode_with_state failed")
            if keep_logits is not None:
                import numpy as np
                nv = w.whisper_n_vocab(ctx)
                keep_logits.append(np.ctypeslib.as_array(w.whisper_get_logits_from_state(st), shape=(nv,)).copy())

    def chunk_all(self, n_decode: int):
        """every stream processes one chunk, concurrently; returns after all are done"""
        errs = []

        def run(s):
            try:
                self.chunk_one(s, n_decode)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=run, args=(s,)) for s in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    def close(self):
        for st in self.states:
            self.w.whisper_free_state(st)
        self.states = []
