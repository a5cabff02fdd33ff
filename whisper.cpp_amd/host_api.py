"""ctypes binding of include/mi355x_host.h (libmi355x_host.so): the native multi-device / multi-stream harness above whisper.h."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import LIB_DIR, PLUGIN_SO, HOST_DIR

HOST_SO = LIB_DIR / "libmi355x_host.so"


class Config(C.Structure):
    _fields_ = [("model_path", C.c_char_p), ("plugin_path", C.c_char_p), ("use_gpu", C.c_int32), ("n_devices", C.c_int32), ("first_device", C.c_int32),
                ("streams_per_device", C.c_int32), ("n_decode", C.c_int32), ("steps", C.c_int32), ("warmup", C.c_int32), ("n_threads", C.c_int32),
                ("skip_payloads", C.c_int32), ("flash_attn", C.c_int32), ("replicas_on_one_device", C.c_int32), ("device_greedy", C.c_int32), ("batching", C.c_int32), ("transport", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("chunks_per_s", C.c_double), ("ms_per_chunk_per_stream", C.c_double), ("load_s", C.c_double),
                ("bcast_bytes", C.c_double), ("bcast_seconds", C.c_double), ("bcast_buffers", C.c_int32), ("bcast_verified", C.c_int32), ("bcast_transport", C.c_int32), ("bcast_ranks", C.c_int32),
                ("bcast_setup_seconds", C.c_double), ("encode_ms", C.c_double), ("decode_ms_per_token", C.c_double),
                ("payload_bytes_read", C.c_int64), ("file_bytes", C.c_int64), ("n_devices", C.c_int32), ("streams_per_device", C.c_int32), ("error", C.c_char * 256), ("greedy_checked", C.c_int64), ("greedy_mismatches", C.c_int64), ("batch_stats", C.c_uint64 * 5)]


TRANSPORTS = {"rccl": 0, "peer": 1, "rccl-world1": 2}
TRANSPORT_NAMES = {0: None, 1: "rccl", 2: "peer", 3: "copy on one device"}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not HOST_SO.exists():
            raise RuntimeError(f"{HOST_SO} is missing: run `python whisper.cpp_amd/build.py host` where the reference tree exists")
        for n in ("libggml-base.so", "libggml-cpu.so", "libggml.so", "libwhisper.so"):
            C.CDLL(str(HOST_DIR / n), mode=C.RTLD_GLOBAL)
        L = C.CDLL(str(HOST_SO))
        L.mi355x_host_run.argtypes = [C.POINTER(Config), C.POINTER(Result)]
        L.mi355x_host_probe_skipping_loader.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
        L.mi355x_host_last_logits.argtypes = [C.c_void_p, C.c_int64]
        L.mi355x_host_open.restype = C.c_void_p
        L.mi355x_host_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def run(model: Path, *, use_gpu: bool, n_devices: int = 1, streams: int = 1, n_decode: int = 256, steps: int = 1, warmup: int = 1,
        n_threads: int = 4, skip_payloads: bool = True, first_device: int = 0, flash_attn: bool = True, replicas_on_one_device: bool = False,
        batching: int = -1, device_greedy: bool = False, transport: str = "rccl") -> dict:
    """transport: "rccl" (default: in-process communicators + grouped ncclBroadcast), "peer" (hipMemcpyPeerAsync), "rccl-world1" (RCCL also with one context)"""
    assert transport in TRANSPORTS, transport
    cfg = Config(str(model).encode(), str(PLUGIN_SO).encode() if use_gpu else None, int(use_gpu), n_devices, first_device, streams, n_decode, steps, warmup,
                 n_threads, int(skip_payloads), int(flash_attn), int(replicas_on_one_device), int(device_greedy), int(batching), TRANSPORTS[transport])
    res = Result()
    rc = lib().mi355x_host_run(C.byref(cfg), C.byref(res))
    d = {f: getattr(res, f) for f, _ in Result._fields_}
    d["error"] = res.error.decode()
    d["batch_stats"] = dict(zip(("chains", "columns", "solo_steps", "fallbacks", "timeouts"), (int(x) for x in res.batch_stats)))
    d["bcast_transport"] = TRANSPORT_NAMES.get(res.bcast_transport)
    d["rc"] = rc
    return d
