"""whisper's log-mel front end on the GPU for hosts that hold PCM: PCM -> HBM -> mi355x_log_mel (csrc/kernels/mel.hip) -> whisper_set_mel.

Reference: log_mel_spectrogram / whisper_pcm_to_mel (src/whisper.cpp:3046-3283, :3901) compute the spectrogram on host threads and hand it to
the encoder through whisper_set_mel's layout (data[j * n_len + i], src/whisper.cpp:2406); the kernel writes exactly that layout.  SURVEY.md §8 row f4.
Device memory comes from the process's HIP runtime through ctypes (no torch): bench.py and the tests pass the handle they already hold.
"""
from __future__ import annotations

import ctypes as C
import struct
import time
from pathlib import Path

import numpy as np


def model_filters(model_path: Path) -> np.ndarray:
    """the mel filterbank of a ggml whisper model file: u32 magic | 11 x i32 hparams | i32 n_mel | i32 n_fft | f32[n_mel][n_fft] (src/whisper.cpp:1485-1560)"""
    with open(model_path, "rb") as f:
        f.seek(4 + 11 * 4)
        n_mel, n_fft = struct.unpack("<ii", f.read(8))
        assert 0 < n_mel <= 256 and n_fft == 201, (n_mel, n_fft)
        return np.frombuffer(f.read(n_mel * n_fft * 4), dtype=np.float32).reshape(n_mel, n_fft).copy()


class GpuFrontEnd:
    def __init__(self, hip, ka, device: int = 0):
        self.hip, self.ka, self.device = hip, ka, device
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipFree.argtypes = [C.c_void_p]
        self._set_device()
        self.ctx = ka.Ctx(device)
        self.bufs = {}                     # role -> (pointer, bytes): reused by later calls, grown when a call needs more

    def _set_device(self):
        if self.hip.hipSetDevice(self.device) != 0:
            raise RuntimeError(f"hipSetDevice({self.device}) failed")

    def _dev(self, role: str, nbytes: int) -> C.c_void_p:
        have = self.bufs.get(role)
        if have and have[1] >= nbytes:
            return have[0]
        if have:
            self.hip.hipFree(have[0])
            del self.bufs[role]
        p = C.c_void_p()
        if self.hip.hipMalloc(C.byref(p), nbytes) != 0:
            raise RuntimeError(f"hipMalloc({nbytes}) failed")
        self.bufs[role] = (p, nbytes)
        return p

    def _copy(self, dst, src, nbytes: int, kind: int, what: str):
        rc = self.hip.hipMemcpy(dst, src, nbytes, kind)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy ({what}, {nbytes} bytes) failed: hipError {rc}")

    def log_mel(self, pcm: np.ndarray, filters: np.ndarray, reps: int = 3):
        """(mel [n_mel, n_len] float32 on the host, best-of-`reps` milliseconds of the device pass: kernel launches + synchronize)"""
        L = self.ka.lib()
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        filters = np.ascontiguousarray(filters, dtype=np.float32)
        n_mel, n_len = filters.shape[0], int(L.mi355x_log_mel_n_len(len(pcm)))
        self._set_device()                                                   # before any allocation: the buffers must live on self.device
        d_pcm, d_f, d_mel = self._dev("pcm", pcm.nbytes), self._dev("filters", filters.nbytes), self._dev("mel", n_mel * n_len * 4)
        self._copy(d_pcm, pcm.ctypes.data, pcm.nbytes, 1, "PCM to device")   # hipMemcpyHostToDevice: the PCM is resident in HBM when the timed pass starts
        self._copy(d_f, filters.ctypes.data, filters.nbytes, 1, "filters to device")
        best = None
        for _ in range(reps):
            self.ctx.sync()
            t0 = time.perf_counter()
            self.ctx.check(L.mi355x_log_mel(self.ctx.h, d_pcm, len(pcm), d_f, n_mel, filters.shape[1], d_mel, n_len), "log_mel")
            self.ctx.sync()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        mel = np.empty((n_mel, n_len), dtype=np.float32)
        self._copy(mel.ctypes.data, d_mel, mel.nbytes, 2, "mel to host")      # hipMemcpyDeviceToHost (whisper_set_mel takes host memory: include/whisper.h)
        return mel, best

    def close(self):
        for p, _ in self.bufs.values():
            self.hip.hipFree(p)
        self.bufs = {}
        self.ctx.close()
