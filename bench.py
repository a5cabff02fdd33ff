#!/usr/bin/env python3
"""whisper-bench encoder+decoder ms per 30 s chunk on MI355X (BASELINE.json metric), N independent streams on N GPUs.

A "step" is one pass of the hot path over one 30 s chunk, with the whisper-bench protocol
(/root/reference/examples/bench/bench.cpp:124-136): 1 x whisper_encode (conv + encoder + cross-KV graphs) followed by
256 x whisper_decode(n_tokens = 1, n_past = i).  The host is the UNMODIFIED reference application
(whisper.cpp_amd/host/_whisper/libwhisper.so, built from the reference sources by oracle/Makefile); all compute runs in
the MI355X ggml backend plugin loaded through ggml's own plugin loader.  GGML_MI355X_STRICT=1 turns any CPU fallback
into an abort, so a number printed here was computed by the HIP kernels.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--arch large-v3] [--qtype q5_0]
  N > 1, one process:   python bench.py --gpus N          N whisper_contexts (device r for context r) in THIS process through the native
                                                           harness (include/mi355x_host.h): replicas r > 0 skip the model file's payloads and
                                                           get their weights by the plugin's peer copy from device 0, checksums compared
  N > 1, one process per GPU (the driver's form):
          python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
                                                           rendezvous / barriers / MAX over ranks on gloo (CPU); the weights travel by the
                                                           plugin's own RCCL broadcast (system librccl).  ONE HIP runtime in every mode: torch
                                                           never touches a GPU here.
  Fewer visible MI355X than --gpus (or than the ranks): the run FAILS with a message, it never shrinks silently.

One JSON line on stdout (rank 0).  `value` = wall ms per chunk aggregated over all streams (T_max / (K*N)); per-stream
latency is `ms_per_step`.  Weights are synthetic (seeded random, real architecture, reference quantizer); mel is seeded
random.  `roofline` comes from a hipEvent-bracketed replay of one extra chunk inside this process; `cpu_baseline` is the
reference CPU path (oracle/_ref, AVX2 build) on a bounded sample, rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy, 79 % of spec) — reported beside the spec peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak
MFMA_I8_PEAK_TOPS = 5000.0     # dense int8 MFMA peak (2 x bf16: 2 x K per instruction); measured micro-benchmark ceiling 4404 TOPS (32x32x32), 3944 (16x16x64) — MI355X_MICROARCH.md
MFMA_I8_MEASURED_TOPS = 4404.0


class ContextParams(C.Structure):       # struct whisper_context_params, include/whisper.h:116-129
    _fields_ = [("use_gpu", C.c_bool), ("flash_attn", C.c_bool), ("gpu_device", C.c_int), ("dtw_token_timestamps", C.c_bool),
                ("dtw_aheads_preset", C.c_int), ("dtw_n_top", C.c_int), ("dtw_n_heads", C.c_size_t), ("dtw_heads", C.c_void_p),
                ("dtw_mem_size", C.c_size_t)]


class Timings(C.Structure):             # struct whisper_timings, include/whisper.h:438-444
    _fields_ = [("sample_ms", C.c_float), ("encode_ms", C.c_float), ("decode_ms", C.c_float), ("batchd_ms", C.c_float), ("prompt_ms", C.c_float)]


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char_p), ("calls", C.c_uint64), ("total_ms", C.c_double), ("algo_bytes", C.c_double), ("algo_flops", C.c_double)]


def load_host(plugin: Path):
    host = ROOT / "whisper.cpp_amd" / "host" / "_whisper"
    if not (host / "libwhisper.so").exists():
        raise RuntimeError(f"{host}/libwhisper.so missing: run `python -c 'import __graft_entry__ as g; g.build()'` where the reference tree exists")
    if not plugin.exists():
        raise RuntimeError(f"{plugin} missing (no CPU fallback exists): build first")
    for n in ("libggml-base.so", "libggml-cpu.so", "libggml.so"):
        C.CDLL(str(host / n), mode=C.RTLD_GLOBAL)
    w = C.CDLL(str(host / "libwhisper.so"), mode=C.RTLD_GLOBAL)
    C.CDLL(str(ROOT / "whisper.cpp_amd" / "lib" / "libmi355x_kernels.so"), mode=C.RTLD_GLOBAL)
    g = C.CDLL(str(host / "libggml.so"))
    g.ggml_backend_load.restype = C.c_void_p
    g.ggml_backend_load.argtypes = [C.c_char_p]
    reg = g.ggml_backend_load(str(plugin).encode())
    if not reg:
        raise RuntimeError("ggml_backend_load rejected the MI355X plugin (no gfx950 device?)")
    p = C.CDLL(str(plugin))
    w.whisper_context_default_params.restype = ContextParams
    w.whisper_init_from_file_with_params.restype = C.c_void_p
    w.whisper_init_from_file_with_params.argtypes = [C.c_char_p, ContextParams]
    w.whisper_set_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    w.whisper_encode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    w.whisper_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    w.whisper_model_n_mels.argtypes = [C.c_void_p]
    w.whisper_get_timings.restype = C.POINTER(Timings)
    w.whisper_get_timings.argtypes = [C.c_void_p]
    w.whisper_reset_timings.argtypes = [C.c_void_p]
    w.whisper_free.argtypes = [C.c_void_p]
    p.ggml_backend_mi355x_prof_report_all.argtypes = [C.POINTER(ProfRow), C.c_int]
    p.ggml_backend_mi355x_stats.argtypes = [C.POINTER(C.c_uint64)]
    return w, p


def broadcast_weights(p, dist, torch, device: int, rank: int, world: int):
    """one-time RCCL broadcast (over xGMI) of rank 0's WEIGHTS buffers into the identically laid out buffers
    of every other replica, done NATIVELY by the plugin (ggml_backend_mi355x_broadcast_weights_rccl: its own communicator from a
    unique id that rank 0 creates and this harness hands to every rank, ncclBroadcast per buffer, then a device-side checksum of
    every buffer compared across all ranks).  Outside the timed region; no collective is ever issued on the per-chunk path.
    Anything but a VERIFIED broadcast raises: replicas r > 0 never read the tensor payloads of the model file."""
    from whisper_cpp_amd.dist_timing import all_ranks_ok, share_bytes
    uid = (C.c_ubyte * 128)()
    got_id = rank != 0 or p.ggml_backend_mi355x_rccl_unique_id(uid) == 0
    if not all_ranks_ok(dist, torch, got_id, "cpu"):
        raise RuntimeError("ggml_backend_mi355x_rccl_unique_id failed on rank 0 (librccl.so not loadable?)")
    uid = (C.c_ubyte * 128)(*share_bytes(dist, torch, bytes(uid) if rank == 0 else None, 128, "cpu"))   # over the harness's own process group
    stats = (C.c_double * 4)()
    p.ggml_backend_mi355x_broadcast_weights_rccl.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
    t0 = time.perf_counter()
    rc = p.ggml_backend_mi355x_broadcast_weights_rccl(device, rank, world, uid, stats)
    wall = time.perf_counter() - t0
    if not all_ranks_ok(dist, torch, rc == 0 and stats[3] == 1.0, "cpu"):
        raise RuntimeError(f"weight broadcast failed or could not be verified on some rank (this rank: rc={rc}, verified={stats[3]})")
    return {"bytes": int(stats[0]), "buffers": int(stats[2]), "seconds": round(stats[1], 4), "GBps": round(stats[0] / max(stats[1], 1e-9) / 1e9, 2),
            "verified": True, "transport": "RCCL ncclBroadcast issued by the plugin (ggml_backend_mi355x_broadcast_weights_rccl), checksums compared across ranks",
            "seconds_incl_communicator_setup": round(wall, 3)}


def algorithmic_figures(arch: str, qtype: str):
    """algorithmic HBM bytes per decoded token and FLOPs per encode, from the hyper-parameters."""
    from whisper_cpp_amd.archs import ARCHS
    (n_vocab, n_actx, n_as, n_ah, n_al, n_tctx, n_ts, n_th, n_tl, n_mels) = ARCHS[arch]
    bpw = {"q4_0": 18 / 32, "q4_k": 144 / 256, "q5_0": 22 / 32, "q8_0": 34 / 32, "f16": 2.0}[qtype]
    n_ctx_pad = (n_actx + 255) // 256 * 256
    dec_w = n_tl * (4 + 4 + 8) * n_ts * n_ts + n_vocab * n_ts            # self-attn 4n^2, cross-attn q/k(v precomputed)/o ... + mlp 8n^2 + logits
    dec_w = n_tl * (4 * n_ts * n_ts + 2 * n_ts * n_ts + 8 * n_ts * n_ts) + n_vocab * n_ts
    kv_cross = n_tl * 2 * n_ctx_pad * n_ts * 2
    gemm_flop = 2.0 * n_actx * n_al * 12 * n_as * n_as + 2.0 * n_actx * n_tl * 2 * n_ts * n_ts          # encoder layers' products + cross-K/V: quantized weights -> int8 MFMA
    f16_flop = 4.0 * n_actx * n_ctx_pad * n_as * n_al + 2.0 * (2 * n_actx) * n_as * 3 * n_mels + 2.0 * n_actx * n_as * 3 * n_as      # attention + the two convolutions: f16 MFMA
    enc_flop = gemm_flop + f16_flop
    # the encoder's / cross-K/V products (1500 columns) run on the f16 matrix cores by default since round 6 (GGML_MI355X_MMQ=1: by width); =2 puts them back on the int8 tile GEMM
    int8 = qtype != "f16" and os.environ.get("GGML_MI355X_MMQ", "1") == "2"
    # roofline time of one encode: every part at the dense peak of the matrix-core type it runs on
    enc_bound_ms = (gemm_flop / ((MFMA_I8_PEAK_TOPS if int8 else MFMA_F16_PEAK_TFLOPS) * 1e12) + f16_flop / (MFMA_F16_PEAK_TFLOPS * 1e12)) * 1e3
    return {"decode_bytes_per_token": dec_w * bpw + kv_cross, "encode_flop": enc_flop, "decode_weight_bytes": dec_w * bpw, "decode_kv_bytes_per_stream": kv_cross,
            "encode_flop_int8": gemm_flop if int8 else 0.0, "encode_flop_f16": f16_flop if int8 else enc_flop, "encode_bound_ms": enc_bound_ms}


DTYPE = ("int8 dot (decode mat-vecs) / int8 MFMA (merged-chain mat-vecs, quantized products with 9..1023 columns) / f16 MFMA on f16(d*q) operands "
         "(encoder and cross-K/V products from 1024 columns on, attention, conv), f32 accumulate")


def contract_line(a, n_gpus: int, streams_per_gpu: int, ms_per_step: float, agg_ms: float, chunks_per_s: float, data="synthetic"):
    """the fields the driver's contract names, in one place (every mode prints the same line shape)"""
    return {"metric": "whisper-bench encoder+decoder ms per 30s chunk", "value": round(agg_ms, 4), "unit": "ms/chunk",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": False,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": data,
            "config": {"workload": f"{a.arch} {a.qtype.upper()}: 1 x whisper_encode + {a.n_decode} x whisper_decode(1 token), {streams_per_gpu} stream{'s' if streams_per_gpu > 1 else ''} per GPU",
                       "streams": n_gpus * streams_per_gpu, "flash_attn": True, "weights": "seeded random, reference quantizer", "mel": "seeded uniform(-1,1)"},
            "chunks_per_s": round(chunks_per_s, 4)}


def n1_cache_path() -> Path:
    return Path(os.environ.get("WHISPER_SYNTH_DIR", "/tmp/whisper_synth")) / "bench_n1_line.json"


def n1_line_of_this_box(a):
    """the N = 1 line this box printed earlier for the same workload (the driver runs N = 1, 2, 4, 8 back to back): its per-kernel roofline object and its
    cpu_baseline are passed through into the N > 1 line — they are properties of one GPU and of the host, not of N"""
    try:
        d = json.loads(n1_cache_path().read_text())
        if d.get("config", {}).get("workload", "").startswith(f"{a.arch} {a.qtype.upper()}:") and time.time() - d.get("_written", 0) < 6 * 3600:
            d["_from"] = f"{n1_cache_path()} written {int(time.time() - d['_written'])} s earlier by `bench.py --gpus 1` on this box"
            return d
    except Exception:  # noqa: BLE001
        pass
    return None


def run_in_process(a, cpu_selftest: bool, hip_runtime: str):
    """--gpus N without torchrun: N whisper_contexts (device r for context r) x --streams states each in THIS process, one host thread per state,
    all started together; wall = the slowest thread (the MAX over GPUs the contract asks for).  Contexts r > 0 open the model file through the
    payload-skipping loader and receive every weight byte by the plugin's device-to-device copy from device 0; every destination buffer's
    checksum must equal device 0's or the run fails (include/mi355x_host.h).  Reference: one context per device
    (include/whisper.h:119, src/whisper.cpp:1297-1311), one state per thread (src/whisper.cpp:7813-7941)."""
    sys.path.insert(0, str(ROOT / "scripts"))
    from synth_model import make_model
    from whisper_cpp_amd import host_api
    model = make_model(a.arch, a.qtype)
    r = host_api.run(model, use_gpu=not cpu_selftest, n_devices=a.gpus, streams=a.streams, n_decode=a.n_decode, steps=a.steps, warmup=a.warmup,
                     skip_payloads=not cpu_selftest, transport=a.transport or "rccl")
    rccl_error = None
    if r["rc"] == 4 and not cpu_selftest and a.transport is None:
        # the DEFAULT transport failed (no usable librccl on this host, or a collective error): the run is repeated on peer copies and the line SAYS so —
        # transport "peer", rccl_error = what happened.  With an explicit --transport rccl the failure is final.
        rccl_error = r["error"]
        print(f"bench.py: RCCL weight broadcast failed ({rccl_error}); repeating with --transport peer", file=sys.stderr)
        r = host_api.run(model, use_gpu=True, n_devices=a.gpus, streams=a.streams, n_decode=a.n_decode, steps=a.steps, warmup=a.warmup, skip_payloads=True, transport="peer")
    if r["rc"] != 0:
        raise SystemExit(f"bench.py: mi355x_host_run failed: rc={r['rc']} {r['error']}")
    if r["n_devices"] != a.gpus:
        raise SystemExit(f"bench.py: the harness ran on {r['n_devices']} devices, {a.gpus} were asked for")
    if not cpu_selftest and r["bcast_verified"] != 1:
        raise SystemExit("bench.py: the weight distribution to the replicas could not be verified: refusing to benchmark replicas with unknown weights")
    if not cpu_selftest and rccl_error is None and (a.transport or "rccl") == "rccl" and not (r["bcast_transport"] == "rccl" and r["bcast_ranks"] == a.gpus):
        raise SystemExit(f"bench.py: asked for an RCCL broadcast over {a.gpus} ranks, the harness reports {r['bcast_transport']} over {r['bcast_ranks']}")
    n_streams = a.gpus * a.streams
    ms_per_step = r["ms_per_chunk_per_stream"]
    out = contract_line(a, a.gpus, a.streams, ms_per_step, ms_per_step / n_streams, r["chunks_per_s"])
    figs = algorithmic_figures(a.arch, a.qtype)
    bound_ms = figs["encode_bound_ms"] + a.n_decode * figs["decode_bytes_per_token"] / (HBM_PEAK_GBS * 1e9) * 1e3
    dec_ms, enc_ms = r["decode_ms_per_token"], r["encode_ms"]
    n1 = n1_line_of_this_box(a)
    out.update({
        "launch": f"one process, {a.gpus} contexts x {a.streams} states, one host thread per state (mi355x_host_run)", "hip_runtime": hip_runtime,
        "encode_ms": round(enc_ms, 3), "decode_ms_per_token": round(dec_ms, 4),
        "weight_broadcast": None if cpu_selftest else {
            "transport": r["bcast_transport"], "ranks": int(r["bcast_ranks"]), "bytes": int(r["bcast_bytes"]), "buffers": int(r["bcast_buffers"]),
            "seconds": round(r["bcast_seconds"], 4), "GBps": round(r["bcast_bytes"] / max(r["bcast_seconds"], 1e-9) / 1e9, 2), "verified": int(r["bcast_verified"]),
            "communicator_setup_seconds": round(r["bcast_setup_seconds"], 3), "rccl_error": rccl_error,
            "how": {"rccl": "ONE communicator per device in this process (ncclCommInitAll), ONE grouped ncclBroadcast per weights buffer from device 0 "
                            "(ggml_backend_mi355x_broadcast_weights_rccl_group), device-side checksums of every buffer of every device compared with device 0's",
                    "peer": "hipMemcpyPeerAsync device 0 -> device r (ggml_backend_mi355x_broadcast_weights_peer, --transport peer), device-side checksums compared"}.get(r["bcast_transport"]),
            "model_file_bytes_read_by_all_contexts": int(r["payload_bytes_read"]), "model_file_bytes": int(r["file_bytes"])},
        # per-GPU figures against the same peaks as the N = 1 line: every GPU runs the same chunk, the means over all streams are per-GPU values.
        # The per-KERNEL roofline needs the hipEvent profile pass, which only the N = 1 form runs: its object is passed through when this box ran it.
        "roofline": {"kernel": "whole chunk (per GPU; per-kernel object: roofline_n1)", "bound": "hbm",
                     "achieved": round(figs["decode_bytes_per_token"] / (dec_ms * 1e-3) / 1e9, 2) if dec_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(figs["decode_bytes_per_token"] / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_ms > 0 else None, "traffic": None,
                     "step_frac": round(figs["decode_bytes_per_token"] / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_ms > 0 else None,
                     "encode_frac": round(figs["encode_bound_ms"] / enc_ms, 4) if enc_ms > 0 else None,
                     "chunk_frac": round(bound_ms / ms_per_step, 4) if ms_per_step > 0 else None},
        "roofline_n1": n1.get("roofline") if n1 else None,
        "cpu_baseline": n1.get("cpu_baseline") if n1 else {"value": None, "unit": "ms/chunk", "cores": None, "kind": "reference",
                                                            "sample": "not measured in the N > 1 form; the N = 1 run of this box leaves it in $WHISPER_SYNTH_DIR/bench_n1_line.json"},
        "n1_line_from": n1.get("_from") if n1 else None,
        "backend": "reference CPU backend: harness self-test, not a measurement" if cpu_selftest else "MI355X plugin, GGML_MI355X_STRICT=1",
    })
    print(json.dumps(out))


def run_ranks_selftest(a, dist, torch, rank: int, world: int):
    """BENCH_BACKEND=cpu under torchrun (tests/test_host.py): the rank protocol of the GPU path — gloo rendezvous, barriers, MAX over ranks, one JSON
    line from rank 0 with n_gpus = WORLD_SIZE — on the reference CPU backend.  Not a measurement."""
    sys.path.insert(0, str(ROOT / "scripts"))
    from synth_model import make_model
    from whisper_cpp_amd import host_api
    from whisper_cpp_amd.dist_timing import aggregate, timed_region
    if rank == 0:
        make_model(a.arch, a.qtype)
    dist.barrier()
    model = make_model(a.arch, a.qtype)
    rd = C.c_int64(0)
    L = host_api.lib()
    ctx = L.mi355x_host_open(str(model).encode(), 0, 0, 1, 0, C.byref(rd))
    if not ctx:
        raise SystemExit("mi355x_host_open failed")
    L.mi355x_host_chunk.argtypes = [C.c_void_p, C.c_int, C.c_int]
    import numpy as np
    w = C.CDLL(str(ROOT / "whisper.cpp_amd" / "host" / "_whisper" / "libwhisper.so"))
    w.whisper_model_n_mels.argtypes = [C.c_void_p]
    w.whisper_set_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    n_mels = w.whisper_model_n_mels(ctx)
    mel = (np.random.default_rng(42 + rank).random((n_mels, 3000), dtype=np.float32) * 2 - 1)
    w.whisper_set_mel(ctx, mel.ctypes.data_as(C.c_void_p), 3000, n_mels)

    def chunk():
        if L.mi355x_host_chunk(ctx, a.n_decode, 2) != 0:
            raise RuntimeError("mi355x_host_chunk failed")
    for _ in range(a.warmup):
        chunk()
    el = timed_region(chunk, a.steps, dist, None, "cpu")
    if rank == 0:
        ms, agg, cps = aggregate(el, a.steps, world)
        out = contract_line(a, world, 1, ms, agg, cps)
        out["backend"] = "reference CPU backend: harness self-test, not a measurement"
        out["launch"] = f"{world} processes (torchrun), gloo rendezvous"
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node (default: 1, or WORLD_SIZE under torchrun)")
    ap.add_argument("--transport", default=None, choices=["rccl", "peer"],
                    help="how replicas r > 0 receive the weights in the one-process form: rccl (default; in-process communicators + grouped ncclBroadcast) or peer "
                         "(hipMemcpyPeerAsync).  Given explicitly with --gpus 1, `rccl` also runs the world-of-one communicator + broadcast + verification and records it")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--arch", default="large-v3")
    ap.add_argument("--qtype", default="q5_0")
    ap.add_argument("--n-decode", type=int, default=256)
    ap.add_argument("--streams", type=int, default=1, help="concurrent streams per GPU in the timed region (headline config: 1)")
    ap.add_argument("--multi-stream", type=int, default=8, help="also measure S concurrent streams on one GPU after the headline, with and without cross-state batching (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="whisper-bench", choices=["whisper-bench", "bounded"],
                    help="whisper-bench: the metric's own tool, the reference build's `whisper-bench -ng -t N` (all four columns, after its own two "
                         "heat-up rounds); bounded: 1 cold encode + 16 decodes through oracle/cpu_baseline.cpp, extrapolated (for very slow hosts)")
    ap.add_argument("--profile-only", action="store_true", help="print the per-kernel hipEvent profile of one chunk and exit")
    ap.add_argument("--profile-what", default="chunk", choices=["chunk", "batchd", "prompt"], help="what --profile-only measures")
    ap.add_argument("--py-loop", action="store_true", help="issue every whisper_decode from Python (ctypes) instead of the native chunk loop")
    ap.add_argument("--no-profile", action="store_true", help="skip the hipEvent per-kernel pass (no roofline object; used under rocprofv3 --pmc)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # also with one process: the distribution path is exercised
    if a.gpus is None:
        a.gpus = world if under_torchrun else 1                               # only an explicitly contradicting --gpus is refused below
    os.environ.setdefault("GGML_MI355X_STRICT", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one hardware queue per concurrent stream (+ the upload stream): with ROCm's default of 4, two of the 4 + 1 HIP streams of
    # the multi-stream pass share a queue and serialize (measured 4.7 -> 5.8 chunks/s at 4 streams; no effect on one stream)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    # ONE HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (ROCm 7.0, no SONAME) while the plugin links the
    # image's /opt/rocm libamdhip64.so.7 (ROCm 7.2); loaded side by side both runtimes initialise the device and every decode step of the
    # plugin gets slower (measured on the same box: 362 vs 348 ms/chunk; the stock whisper-bench process, which has only the system runtime,
    # 339).  Bringing the system runtime into the global symbol scope BEFORE torch is imported makes torch's HIP calls resolve to it as well
    # (what LD_PRELOAD does).
    # BENCH_SYSTEM_HIP=0 switches this off, =1 forces it.
    sys_hip = os.environ.get("BENCH_SYSTEM_HIP", "1")      # (under torchrun too since r05: the process group is gloo, torch's RCCL is never used)
    hip_runtime = "pytorch wheel's libamdhip64 next to the system one"
    if sys_hip == "1":
        for cand in ("/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
            if os.path.exists(cand):
                try:
                    C.CDLL(cand, mode=C.RTLD_GLOBAL)
                    hip_runtime = cand + " for the whole process"
                except OSError:
                    pass
                break

    import numpy as np
    import __graft_entry__ as graft
    graft.load_package()
    cpu_selftest = os.environ.get("BENCH_BACKEND", "gpu") == "cpu"          # harness self-test on the reference CPU backend (tests/test_host.py): NOT a measurement
    n_visible = 0
    if not cpu_selftest:
        from whisper_cpp_amd import kernels_api as _ka
        n_visible = int(_ka.lib().mi355x_device_count())
        if n_visible < 1:
            raise SystemExit("bench.py needs an MI355X: no gfx950 device is visible (there is no CPU fallback)")
    if under_torchrun and a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE is {world}: launch one rank per GPU")
    if not cpu_selftest and (local_rank >= n_visible or (not under_torchrun and a.gpus > n_visible)):
        raise SystemExit(f"bench.py: --gpus {a.gpus} (local rank {local_rank}) but only {n_visible} MI355X visible: refusing to run on fewer GPUs than asked for")
    if a.gpus > 1 and not under_torchrun:
        return run_in_process(a, cpu_selftest, hip_runtime)

    import torch          # CPU tensors of the gloo process group only: no torch.cuda call in this file
    dist = None
    if under_torchrun:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("gloo")
        if cpu_selftest:
            return run_ranks_selftest(a, dist, torch, rank, world)
    elif cpu_selftest:
        raise SystemExit("bench.py: BENCH_BACKEND=cpu is the multi-GPU harness self-test: use --gpus N > 1 or torchrun")

    hip = None
    if not cpu_selftest:
        hip = C.CDLL("/opt/rocm/lib/libamdhip64.so.7" if os.path.exists("/opt/rocm/lib/libamdhip64.so.7") else "/opt/rocm/lib/libamdhip64.so")

    def device_sync():
        if hip is not None:
            hip.hipSetDevice(local_rank)
            if hip.hipDeviceSynchronize() != 0:
                raise RuntimeError("hipDeviceSynchronize failed")

    def barrier():
        if dist is not None:
            dist.barrier()

    sys.path.insert(0, str(ROOT / "scripts"))
    from synth_model import make_model          # tooling: writes the synthetic model file with the reference application's own quantizer

    # rank 0 writes the synthetic model file once; every rank loads the same file (one replica per GPU).
    # Replicas are independent streams: no data-path collective.
    if rank == 0:
        model = make_model(a.arch, a.qtype)
    barrier()
    model = make_model(a.arch, a.qtype)

    plugin = ROOT / "whisper.cpp_amd" / "lib" / "libggml-mi355x.so"
    w, p = load_host(plugin)
    bcast, payload_read = None, None
    if dist is None:
        cp = w.whisper_context_default_params()
        cp.use_gpu, cp.flash_attn, cp.gpu_device = True, True, local_rank
        ctx = w.whisper_init_from_file_with_params(str(model).encode(), cp)
        if not ctx:
            raise SystemExit("whisper_init_from_file_with_params failed")
    else:
        # replicas r > 0 open the model through the payload-skipping whisper_model_loader of the native host library (header,
        # filters, vocabulary only) and receive every weight byte from rank 0's HBM; an unverified broadcast ends the run (rc != 0)
        from whisper_cpp_amd import host_api
        rd = C.c_int64(0)
        ctx = host_api.lib().mi355x_host_open(str(model).encode(), 1, local_rank, 1, 1 if rank > 0 else 0, C.byref(rd))
        if not ctx:
            raise SystemExit("mi355x_host_open failed")
        try:
            bcast = broadcast_weights(p, dist, torch, local_rank, rank, world)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}: {e}", file=sys.stderr)
            os._exit(3)
        bcast["model_file_bytes_read_by_this_rank"] = int(rd.value)
        bcast["model_file_bytes"] = Path(model).stat().st_size
    n_mels = w.whisper_model_n_mels(ctx)
    mel = (np.random.default_rng(42 + rank).random((n_mels, 3000), dtype=np.float32) * 2 - 1)
    w.whisper_set_mel(ctx, mel.ctypes.data_as(C.c_void_p), 3000, n_mels)
    tokens = (C.c_int32 * 512)()
    n_threads = 4

    from whisper_cpp_amd.streams import Streams

    def make_streams(n):
        mels = [(np.random.default_rng(1000 * rank + 42 + i).random((n_mels, 3000), dtype=np.float32) * 2 - 1) for i in range(n)]
        return Streams(w, ctx, n, mels)

    multi = make_streams(a.streams) if a.streams > 1 else None
    from whisper_cpp_amd import host_api as _host_api
    native_chunk = _host_api.lib().mi355x_host_chunk
    native_chunk.argtypes = [C.c_void_p, C.c_int, C.c_int]

    def chunk():
        if multi is not None:                 # --streams S: S states on this GPU, one host thread each, all process one chunk
            multi.chunk_all(a.n_decode)
            return
        if not a.py_loop:                     # the bench protocol's loop in native code (include/mi355x_host.h): no Python call per token
            rc = native_chunk(ctx, a.n_decode, n_threads)
            if rc != 0:
                raise RuntimeError(f"mi355x_host_chunk failed: rc={rc}")
            return
        if w.whisper_encode(ctx, 0, n_threads) != 0:
            raise RuntimeError("whisper_encode failed")
        for i in range(a.n_decode):
            if w.whisper_decode(ctx, tokens, 1, i, n_threads) != 0:
                raise RuntimeError("whisper_decode failed")

    def profile_chunk():
        p.ggml_backend_mi355x_prof_enable_all(1)
        p.ggml_backend_mi355x_prof_reset_all()
        if a.profile_what == "batchd":        # the beam-search shape: 5 tokens per step
            for _ in range(16):
                w.whisper_decode(ctx, tokens, 5, 0, n_threads)
        elif a.profile_what == "prompt":
            for _ in range(4):
                w.whisper_decode(ctx, tokens, 256, 0, n_threads)
        else:
            chunk()
        rows = (ProfRow * 64)()
        n = p.ggml_backend_mi355x_prof_report_all(rows, 64)
        p.ggml_backend_mi355x_prof_enable_all(0)
        return [dict(name=rows[i].name.decode(), calls=int(rows[i].calls), total_ms=rows[i].total_ms, algo_bytes=rows[i].algo_bytes,
                     algo_flops=rows[i].algo_flops) for i in range(n)]

    if a.profile_only:
        chunk()
        prof = sorted(profile_chunk(), key=lambda r: -r["total_ms"])
        print(json.dumps({"arch": a.arch, "qtype": a.qtype, "kernels": prof}, indent=1))
        return

    from whisper_cpp_amd.dist_timing import aggregate, timed_region
    for _ in range(a.warmup):
        chunk()
    w.whisper_reset_timings(ctx)
    p.ggml_backend_mi355x_host_times.argtypes = [C.POINTER(C.c_double)]
    host0 = (C.c_double * 13)()
    p.ggml_backend_mi355x_host_times(host0)
    # barrier + synchronize on both sides, exactly K steps, MAX over ranks.  Every whisper_encode / whisper_decode
    # returns only after the backend's stream is drained (ggml_backend_sched_synchronize); hipDeviceSynchronize (system runtime, the
    # plugin's own) additionally drains the device.
    elapsed_s = timed_region(chunk, a.steps, dist, device_sync, "cpu")
    tm = w.whisper_get_timings(ctx).contents
    encode_ms, decode_ms = float(tm.encode_ms), float(tm.decode_ms)
    host1 = (C.c_double * 13)()
    p.ggml_backend_mi355x_host_times(host1)
    host_ms = [host1[i] - host0[i] for i in range(13)]          # host-side time inside the backend during the timed region only

    # reported beside the headline (bench.cpp:138-152): 5-token batches and 256-token prompts
    # (one untimed pass of each shape first, as whisper-bench's own heat-up does, bench.cpp:94-121: the first 256-column pass
    # makes the one-time f16 copies of the decoder weights)
    w.whisper_decode(ctx, tokens, 5, 0, n_threads)
    w.whisper_decode(ctx, tokens, 256, 0, n_threads)
    w.whisper_reset_timings(ctx)
    for _ in range(16):
        w.whisper_decode(ctx, tokens, 5, 0, n_threads)
    for _ in range(4):
        w.whisper_decode(ctx, tokens, 256, 0, n_threads)
    tm = w.whisper_get_timings(ctx).contents
    batchd_ms, prompt_ms = float(tm.batchd_ms), float(tm.prompt_ms)

    # SURVEY.md §8f rank 2, reported beside the headline: S concurrent streams sharing this GPU and one copy of the weights, through the
    # native C++ harness (include/mi355x_host.h: one thread per whisper_state).  "batched": the plugin runs the states' single-token
    # steps as the columns of ONE launch chain (weights read once per step for all streams); "unbatched": one launch chain per state.
    multi_stream = None
    if a.multi_stream > 1 and multi is None and world == 1:
        try:
            from whisper_cpp_amd import host_api
            figs_ms = algorithmic_figures(a.arch, a.qtype)
            multi_stream = {"streams": a.multi_stream, "harness": "mi355x_host_run (C++ threads, one whisper_state each, one whisper_context, weights shared)"}
            # (own chains first: the merged leg creates the plugin's lane streams, which then share hardware queues with the states' streams)
            # own chains at 4 streams (their best point) AND at the batched leg's stream count, so that the two forms can be compared
            legs = [(f"own_chains_{min(a.multi_stream, 4)}_streams", 0, min(a.multi_stream, 4))]
            if a.multi_stream > 4:
                legs.append((f"own_chains_{a.multi_stream}_streams", 0, a.multi_stream))
            legs.append((f"batched_{a.multi_stream}_streams", 1, a.multi_stream))
            if a.multi_stream == 8:
                legs.append(("batched_16_streams", 1, 16))        # chains of 10 + 6 columns per step: from 9 columns on the mat-vecs run on the matrix cores (decode_mx.hip)
                legs.append(("batched_32_streams", 1, 32))        # chains of 20 + 12 columns
                legs.append(("batched_64_streams", 1, 64))        # more states than one chain carries: chains of equal width, 32 + 32 (round 6)
            for label, batching, ns in legs:
                r = host_api.run(model, use_gpu=True, n_devices=1, streams=ns, n_decode=a.n_decode, steps=2, warmup=1, batching=batching)
                if r["rc"] != 0:
                    multi_stream[label] = {"streams": ns, "error": r["error"]}
                    continue
                e = {"streams": ns, "chunks_per_s": round(r["chunks_per_s"], 4), "ms_per_chunk_aggregate": round(1e3 / r["chunks_per_s"], 3),
                     "ms_per_chunk_per_stream": round(r["ms_per_chunk_per_stream"], 3), "batch_stats": r["batch_stats"]}
                if batching:
                    # algorithmic HBM bytes of one merged step: the decoder weights ONCE + every stream's cross-KV; streams * n_decode tokens per chunk round
                    st = r["batch_stats"]
                    cols = st["columns"] / max(st["chains"], 1)
                    step_bytes = figs_ms["decode_weight_bytes"] + cols * figs_ms["decode_kv_bytes_per_stream"]
                    steps_per_s = r["chunks_per_s"] * a.n_decode / max(cols, 1)
                    e["mean_columns_per_chain"] = round(cols, 2)
                    e["decode_algorithmic_GBps"] = round(step_bytes * steps_per_s / 1e9, 1)
                    e["decode_frac_of_hbm_peak"] = round(step_bytes * steps_per_s / 1e9 / HBM_PEAK_GBS, 4)
                multi_stream[label] = e
        except Exception as e:  # noqa: BLE001
            multi_stream = {"streams": a.multi_stream, "error": str(e)}

    # SURVEY.md §8 row f4, reported beside the headline (the headline itself keeps BASELINE.json's synthetic mel input): 30 s of synthetic PCM resident
    # in HBM -> mi355x_log_mel -> whisper_set_mel -> one chunk on that spectrogram.  Reference: whisper_pcm_to_mel on host threads (src/whisper.cpp:3901).
    front_end = None
    if rank == 0 and world == 1 and hip is not None:
        try:
            from whisper_cpp_amd.front_end import GpuFrontEnd, model_filters
            from whisper_cpp_amd import kernels_api as ka_
            t = np.arange(16000 * 30, dtype=np.float64) / 16000.0
            pcm = (0.3 * np.sin(2 * np.pi * (200 + 80 * np.sin(2 * np.pi * 0.5 * t)) * t) + 0.05 * np.random.default_rng(5).standard_normal(t.size)).astype(np.float32)
            fe = GpuFrontEnd(hip, ka_, local_rank)
            mel_gpu, mel_ms = fe.log_mel(pcm, model_filters(Path(model)))
            fe.close()
            if w.whisper_set_mel(ctx, mel_gpu.ctypes.data_as(C.c_void_p), mel_gpu.shape[1], mel_gpu.shape[0]) != 0:
                raise RuntimeError("whisper_set_mel rejected the GPU spectrogram")
            t0 = time.perf_counter()
            chunk()
            front_end = {"mel_ms": round(mel_ms, 3), "n_samples": int(pcm.size), "n_mel": int(mel_gpu.shape[0]), "n_len": int(mel_gpu.shape[1]),
                         "chunk_ms_on_that_mel": round((time.perf_counter() - t0) * 1e3, 2),
                         "path": "PCM in HBM -> mi355x_log_mel (mel.hip) -> whisper_set_mel -> encode + decode; parity: tests/test_gpu.py::test_gpu_log_mel_matches_the_reference_front_end_on_real_speech"}
            w.whisper_set_mel(ctx, mel.ctypes.data_as(C.c_void_p), 3000, n_mels)          # back to the benchmark's input
        except Exception as e:  # noqa: BLE001
            front_end = {"error": str(e)}

    stats = (C.c_uint64 * 4)()
    p.ggml_backend_mi355x_stats(stats)
    prof = profile_chunk() if (rank == 0 and not a.no_profile) else []

    if rank == 0:
        figs = algorithmic_figures(a.arch, a.qtype)
        ms_per_step, agg_ms, chunks_per_s = aggregate(elapsed_s, a.steps, world * a.streams)
        out = contract_line(a, world, a.streams, ms_per_step, agg_ms, chunks_per_s)
        out.update({
            "encode_ms": round(encode_ms, 3), "decode_ms_per_token": round(decode_ms, 4),
            "batchd_ms_per_token": round(batchd_ms, 4), "prompt_ms_per_token": round(prompt_ms, 4),
            "weight_broadcast": bcast, "multi_stream": multi_stream, "front_end": front_end,
            "launch_mode": "plain launches on the backend's stream", "hip_runtime": hip_runtime,
            "backend": {"graph_computes": int(stats[0]),
                        "host_ms_in_timed_region": {"graph_compute": round(host_ms[3], 2),
                                            "set_tensor": round(host_ms[4], 2), "get_tensor": round(host_ms[5], 2), "cpy_tensor": round(host_ms[6], 2), "synchronize": round(host_ms[7], 2),
                                            "calls": [int(host_ms[8 + i]) for i in range(4)], "gpu_span": round(host_ms[12], 2)}},
            "launch": f"{world} process{'es (torchrun), gloo rendezvous, one rank per GPU' if under_torchrun else ''}",
        })
        if prof:
            # the dominant kernel = the kernel TEMPLATE with the most GPU time (its instantiations are one kernel built for different
            # shapes: the decode mat-vec's five share of ~70 % is what bounds the chunk), no tie-break; achieved = the family's summed
            # algorithmic bytes / its summed time, i.e. the time-weighted figure; the single largest instantiation is named beside it
            total = sum(r["total_ms"] for r in prof)
            fam = {}
            for r in prof:
                key = r["name"].split("<")[0].strip()
                f = fam.setdefault(key, {"name": key, "total_ms": 0.0, "calls": 0, "algo_bytes": 0.0, "algo_flops": 0.0, "members": []})
                f["total_ms"] += r["total_ms"]; f["calls"] += r["calls"]; f["algo_bytes"] += r["algo_bytes"]; f["algo_flops"] += r["algo_flops"]
                f["members"].append(r)
            dom = max(fam.values(), key=lambda f: f["total_ms"])
            big = max(dom["members"], key=lambda r: r["total_ms"])
            avg_ms = dom["total_ms"] / max(dom["calls"], 1)
            is_i8 = "mmq" in dom["name"]
            if "gemm" in dom["name"] or "fattn_mfma" in dom["name"] or is_i8:
                ach = dom["algo_flops"] / (dom["total_ms"] * 1e-3) / 1e12
                peak = MFMA_I8_PEAK_TOPS if is_i8 else MFMA_F16_PEAK_TFLOPS          # an int8 kernel is scored against the int8 ceiling
                out["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TOP/s" if is_i8 else "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": None}
                if is_i8:
                    out["roofline"]["measured_ceiling"] = MFMA_I8_MEASURED_TOPS
            else:
                ach = dom["algo_bytes"] / (dom["total_ms"] * 1e-3) / 1e9
                out["roofline"] = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                   "measured_copy_ceiling": HBM_COPY_CEILING_GBS, "frac_of_copy_ceiling": round(ach / HBM_COPY_CEILING_GBS, 4)}
            # HBM traffic from the committed PMC passes of THIS configuration (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md
            # prescribes; profiles/pmc_traffic.json says how it was collected).  Like `achieved`, `traffic` is the FAMILY's figure per launch: the
            # launch-weighted mean over its instantiations of the counter bytes, next to `algorithmic_per_launch`, the same mean of the algorithmic
            # bytes — and every instantiation is listed with its own pair (same kernel, same launch), so that traffic / algorithmic means something.
            pmc = {}
            try:
                pj = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
                pmc = pj.get("configs", {}).get(f"{a.arch} {a.qtype}", {}).get("kernels", {})
            except Exception:  # noqa: BLE001
                pass
            hbm = out["roofline"]["bound"] == "hbm"
            inst = []
            for r in sorted(dom["members"], key=lambda r: -r["total_ms"]):
                key = r["name"].split("(")[0].strip()
                e = {"name": r["name"], "launches": r["calls"], "avg_launch_us": round(r["total_ms"] * 1e3 / max(r["calls"], 1), 3),
                     "algorithmic_per_launch": (r["algo_bytes"] if hbm else r["algo_flops"]) / max(r["calls"], 1),
                     "traffic": pmc[key]["hbm_bytes_per_launch"] if key in pmc else None}
                if hbm:
                    e["GBps"] = round(r["algo_bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1) if r["total_ms"] > 0 else None
                    e["traffic_over_algorithmic"] = round(e["traffic"] / e["algorithmic_per_launch"], 3) if e["traffic"] and e["algorithmic_per_launch"] else None
                inst.append(e)
            if inst and all(e["traffic"] is not None for e in inst):
                out["roofline"]["traffic"] = round(sum(e["traffic"] * e["launches"] for e in inst) / max(dom["calls"], 1))
            out["roofline"].update({"launches": dom["calls"], "avg_launch_us": round(avg_ms * 1e3, 3), "share_of_gpu_time": round(dom["total_ms"] / total, 4),
                                    "algorithmic_per_launch": (dom["algo_bytes"] if hbm else dom["algo_flops"]) / max(dom["calls"], 1),
                                    "instantiations": inst})
            # the whole step against the same peaks: algorithmic bytes per token / measured ms per token / HBM peak, encoder FLOP / measured ms / MFMA peak
            dec_b = figs["decode_bytes_per_token"]
            if decode_ms > 0:
                out["roofline"]["step_frac"] = round(dec_b / (decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if encode_ms > 0:
                # encoder: roofline time (int8 products at the int8 peak, attention / conv at the f16 peak) / measured time
                out["roofline"]["encode_frac"] = round(figs["encode_bound_ms"] / encode_ms, 4)
                out["roofline"]["encode_frac_if_all_f16_peak"] = round(figs["encode_flop"] / (encode_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
            # roofline time of the chunk (encoder at the MFMA peak + n_decode tokens at the HBM peak) / measured time
            bound_ms = figs["encode_bound_ms"] + a.n_decode * dec_b / (HBM_PEAK_GBS * 1e9) * 1e3
            out["roofline"]["chunk_frac"] = round(bound_ms / agg_ms, 4) if agg_ms > 0 else None
            out["kernel_time_ms_per_chunk"] = {r["name"]: round(r["total_ms"], 3) for r in sorted(prof, key=lambda r: -r["total_ms"])}
            # every kernel with >= 2 % of the GPU time: launches, mean duration, achieved algorithmic GB/s and TFLOP/s
            out["kernels"] = [{"name": r["name"], "launches": r["calls"], "avg_us": round(r["total_ms"] * 1e3 / max(r["calls"], 1), 3),
                               "share": round(r["total_ms"] / total, 4),
                               "GBps": round(r["algo_bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1) if r["total_ms"] > 0 else None,
                               "TFLOPs": round(r["algo_flops"] / (r["total_ms"] * 1e-3) / 1e12, 2) if r["total_ms"] > 0 else None}
                              for r in sorted(prof, key=lambda r: -r["total_ms"]) if r["total_ms"] >= 0.02 * total]
            dec_bytes = figs["decode_bytes_per_token"]
            out["step_roofline"] = {"decode_algorithmic_MB_per_token": round(dec_bytes / 1e6, 2),
                                    "decode_GBps_at_measured_ms": round(dec_bytes / (decode_ms * 1e-3) / 1e9, 1) if decode_ms > 0 else None,
                                    "encode_TFLOP": round(figs["encode_flop"] / 1e12, 3),
                                    "encode_TFLOPs_at_measured_ms": round(figs["encode_flop"] / (encode_ms * 1e-3) / 1e12, 2) if encode_ms > 0 else None}
        if world == 1 and not a.no_cpu_baseline:
            # threads actually used: ggml's CPU path stops scaling (and with 2-way SMT oversubscription collapses) well
            # below the 256 hardware threads of the GPU box's host; 32 = one thread per core of half a socket
            cores = max(1, min(32, (os.cpu_count() or 2) // 2))
            env = dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "oracle" / "_ref"), OMP_PROC_BIND="close", OMP_PLACES="cores")
            env.pop("GGML_BACKEND_PATH", None)
            try:
                if a.cpu_baseline == "whisper-bench":
                    # the metric's own tool from the reference build (oracle/_ref, AVX2-only flags: oracle/Makefile), CPU only (-ng),
                    # its own protocol: two heat-up rounds, then 1 encode, 256 x 1-token, 64 x 5-token, 16 x 256-token decodes
                    # (examples/bench/bench.cpp:63-170; scripts/bench-all.sh:73-77 parses the same lines)
                    import re
                    t0 = time.perf_counter()
                    r = subprocess.run([str(ROOT / "oracle" / "_ref" / "whisper-bench"), "-m", str(model), "-ng", "-t", str(cores)], env=env,
                                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
                    wall = time.perf_counter() - t0
                    col = {}
                    for name in ("encode", "decode", "batchd", "prompt"):
                        mm = re.search(rf"{name} time =\s*([0-9.]+) ms /\s*(\d+) runs \(\s*([0-9.]+) ms per run\)", r.stdout)
                        col[name] = (float(mm.group(1)), int(mm.group(2)), float(mm.group(3)))
                    si = re.search(r"system_info: (.*)", r.stdout)
                    enc_ms, dec_ms = col["encode"][2], col["decode"][0] / col["decode"][1]
                    out["cpu_baseline"] = {"value": round(enc_ms + a.n_decode * dec_ms, 2), "unit": "ms/chunk", "cores": cores, "kind": "reference",
                                           "encode_ms": enc_ms, "decode_ms_per_token": round(dec_ms, 4),
                                           "batchd_ms_per_token": round(col["batchd"][0] / col["batchd"][1], 4), "prompt_ms_per_token": round(col["prompt"][0] / col["prompt"][1], 4),
                                           "sample": f"oracle/_ref/whisper-bench -ng -t {cores} (reference build, AVX2-only flags, flash-attn on): warm, after the tool's own two heat-up "
                                                     f"rounds; encode + {a.n_decode} x decode from its 'encode time' and 'decode time' lines; {wall:.0f} s of host wall time",
                                           "system_info": si.group(1).strip() if si else None}
                else:
                    exe = ROOT / "oracle" / "_ref" / "cpu_baseline"
                    n_dec = 16 if "large" in a.arch else 64
                    r = subprocess.run([str(exe), str(model), str(cores), str(n_dec), "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=240)
                    cb = json.loads(r.stdout.strip().splitlines()[-1])
                    out["cpu_baseline"] = {"value": round(cb["encode_ms"] + a.n_decode * cb["decode_ms_per_token"], 2), "unit": "ms/chunk", "cores": cores,
                                           "kind": "reference", "encode_ms": cb["encode_ms"], "decode_ms_per_token": cb["decode_ms_per_token"],
                                           "sample": f"reference AVX2 CPU path (oracle/_ref, use_gpu=false): 1 cold whisper_encode + {n_dec} single-token decodes, "
                                                     f"extrapolated to encode + {a.n_decode} x decode", "system_info": cb["system_info"].strip()}
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "ms/chunk", "cores": cores, "kind": "reference", "sample": f"failed: {e}"}
        if world == 1 and a.transport == "rccl" and bcast is None:
            # --gpus 1 --transport rccl: what one GPU can show of the N > 1 distribution path — a world of one through the same entry point
            try:
                from whisper_cpp_amd import host_api
                rr = host_api.run(model, use_gpu=True, n_devices=1, streams=1, n_decode=4, steps=1, warmup=0, transport="rccl-world1")
                out["weight_broadcast"] = {"transport": rr["bcast_transport"], "ranks": int(rr["bcast_ranks"]), "bytes": int(rr["bcast_bytes"]), "buffers": int(rr["bcast_buffers"]),
                                           "seconds": round(rr["bcast_seconds"], 4), "verified": int(rr["bcast_verified"]), "communicator_setup_seconds": round(rr["bcast_setup_seconds"], 3),
                                           "rc": rr["rc"], "error": rr["error"], "note": "world of one: communicator, grouped broadcast and checksum verification ran; nothing moved"}
            except Exception as e:  # noqa: BLE001
                out["weight_broadcast"] = {"transport": "rccl", "error": str(e)}
        if world == 1 and a.streams == 1:
            try:
                n1_cache_path().write_text(json.dumps(dict(out, _written=time.time())))
            except Exception:  # noqa: BLE001
                pass
        print(json.dumps(out))
    barrier()                   # rank 0 may still be profiling / printing: nobody tears the process group down under it
    w.whisper_free(ctx)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
