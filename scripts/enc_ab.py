#!/usr/bin/env python3
"""Whole-encode A-B in ONE process: the model is loaded once through the unmodified libwhisper + plugin, then `whisper_encode` is timed
(whisper's own timers) under each setting of the kernel library's run-time switches (environment variables read at every launch:
GGML_MI355X_FATTN_NG, GGML_MI355X_GEMM_RING_TM256[_MIN], GGML_MI355X_GEMM_GROUP_CFG).  Per setting: encode ms (mean of --reps after 2
untimed passes) and the per-kernel totals of one profiled encode.

  python scripts/enc_ab.py [--arch large-v3] [--qtype q5_0] [--reps 8] -- "VAR=1 OTHER=2" "" ...
An empty string is the default configuration."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="large-v3")
    ap.add_argument("--qtype", default="q5_0")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("cfgs", nargs="*")
    a = ap.parse_args()
    os.environ.setdefault("GGML_MI355X_STRICT", "1")
    for cand in ("/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            break
    import numpy as np
    import bench
    import __graft_entry__ as graft
    graft.load_package()
    from synth_model import make_model
    model = make_model(a.arch, a.qtype)
    w, p = bench.load_host(ROOT / "whisper.cpp_amd" / "lib" / "libggml-mi355x.so")
    cp = w.whisper_context_default_params()
    cp.use_gpu, cp.flash_attn, cp.gpu_device = True, True, 0
    ctx = w.whisper_init_from_file_with_params(str(model).encode(), cp)
    if not ctx:
        raise SystemExit("whisper_init_from_file_with_params failed")
    n_mels = w.whisper_model_n_mels(ctx)
    mel = np.random.default_rng(42).random((n_mels, 3000), dtype=np.float32) * 2 - 1
    w.whisper_set_mel(ctx, mel.ctypes.data_as(C.c_void_p), 3000, n_mels)
    for _ in range(2):
        w.whisper_encode(ctx, 0, 4)
    print(f"# {a.arch} {a.qtype}: whisper_encode ms (mean of {a.reps}) per setting of the kernel library's switches; kernel totals of one profiled encode, ms")
    for cfg in a.cfgs or [""]:
        kv = dict(x.split("=", 1) for x in cfg.split()) if cfg.strip() else {}
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        try:
            for _ in range(2):
                if w.whisper_encode(ctx, 0, 4) != 0:
                    raise RuntimeError("whisper_encode failed")
            w.whisper_reset_timings(ctx)
            for _ in range(a.reps):
                w.whisper_encode(ctx, 0, 4)
            enc = float(w.whisper_get_timings(ctx).contents.encode_ms)
            p.ggml_backend_mi355x_prof_enable_all(1)
            p.ggml_backend_mi355x_prof_reset_all()
            w.whisper_encode(ctx, 0, 4)
            rows = (bench.ProfRow * 64)()
            n = p.ggml_backend_mi355x_prof_report_all(rows, 64)
            p.ggml_backend_mi355x_prof_enable_all(0)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        tot = {}
        for i in range(n):
            nm = rows[i].name.decode()
            key = ("group" if "ring_group" in nm else "ring" if "k_gemm_f16_ring" in nm else "fattn" if "k_fattn_mfma" in nm else
                   "norm" if "k_norm" in nm else "other")
            tot[key] = tot.get(key, 0.0) + rows[i].total_ms
        ks = " ".join(f"{k} {tot.get(k, 0):.3f}" for k in ("group", "ring", "fattn", "norm", "other"))
        print(f"{cfg or '(default)':64s} encode {enc:7.3f} ms   {ks}   sum {sum(tot.values()):.3f}", flush=True)
    w.whisper_free(ctx)


if __name__ == "__main__":
    main()
