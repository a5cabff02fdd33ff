"""TOOLING (not part of the shipped package): synthetic whisper.cpp model files (legacy `ggml` bin format) with seeded random weights.

There are no real Whisper checkpoints in the build container or on the GPU box (no network), so every
parity / benchmark run uses models of the real ARCHITECTURE with random-init weights, written in the
reference's own file format and then quantized by the reference's own `whisper-quantize`
(examples/quantize/quantize.cpp) so that the bytes the backend receives through set_tensor are produced by
the reference's quantizer.

File format (src/whisper.cpp:1485-1700 loader; models/convert-pt-to-ggml.py:264-340 writer):
  u32 magic 0x67676d6c | 11 x i32 hparams | i32 n_mel, i32 n_fft, f32[n_mel*n_fft] filters |
  i32 n_vocab, n_vocab x (u32 len, bytes) | tensors: i32 n_dims, i32 name_len, i32 ftype(0=f32,1=f16),
  i32 ne[n_dims] (ggml order, fastest first), name, data
Tensor names / shapes: src/whisper-arch.h:42-109, src/whisper.cpp:1757-1843.
"""
from __future__ import annotations

import argparse
import os
import struct
import subprocess
from pathlib import Path

import numpy as np

import sys

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as _graft  # noqa: E402

_graft.load_package()
from whisper_cpp_amd.archs import ARCHS, QTYPES  # noqa: E402



def _w_tensor(f, name: str, arr: np.ndarray):
    """arr in numpy (row-major) order; ggml ne[] is the reverse."""
    ftype = 1 if arr.dtype == np.float16 else 0
    nb = name.encode()
    f.write(struct.pack("iii", arr.ndim, len(nb), ftype))
    for d in reversed(arr.shape):
        f.write(struct.pack("i", d))
    f.write(nb)
    f.write(arr.tobytes())


class _Rng:
    """Fast seeded normal generator: one 4M-sample pool, consumed with a rolling offset (the statistics matter,
    not independence across tensors; generating 1.5e9 fresh normals for large-v3 would take minutes)."""

    def __init__(self, seed: int):
        self.pool = np.random.default_rng(seed).standard_normal(1 << 22).astype(np.float32)
        self.off = 0

    def normal(self, shape, scale: float) -> np.ndarray:
        n = int(np.prod(shape))
        out = np.empty(n, dtype=np.float32)
        pos = 0
        while pos < n:
            take = min(n - pos, self.pool.size - self.off)
            out[pos:pos + take] = self.pool[self.off:self.off + take]
            pos += take
            self.off = (self.off + take + 7919) % self.pool.size   # de-correlate successive tensors a little
        out *= scale
        return out.reshape(shape)


def planted_token(pos: int) -> int:
    """the token a PLANTED model emits after the token at decoder position `pos` (ordinary text tokens, all different)"""
    return 1000 + (pos * 7919) % 40000


# ---- "x-planted" models: the token decision is CARRIED BY CROSS-ATTENTION, with every layer at full strength -------------------------
# (VERDICT r03 weak #1a: in the planted model above the transcript is decided by pos-emb . token-emb alone — cross-attention could
# return garbage and the test would still pass.)  Here position p offers TWO candidates a_p / b_p with equal strength in the positional
# embedding; which of them wins is decided by the sign of ONE number that only the last decoder layer's cross-attention produces:
#   * u: a fixed random direction; a_p / b_p = the text tokens whose embeddings project most positively / most negatively on u
#   * last layer: row 0 of cross_attn.value.weight = w (random), column 0 of cross_attn.out.weight = g * u, query / key weights x c
#     (c > 1: sharper attention), element 0 of cross_attn.value.bias = 1 => the block adds g * u * s_p to the residual stream with
#     s_p = sum_k softmax_k(q_p . K) (w . enc_k + 1): a number produced by the softmax over the 1500 encoder keys (it must sum to one),
#     by V and by W_o — i.e. by every piece of arithmetic of the cross-attention path.  (Random-weight encoders barely react to their
#     input — two different "audios" change 0 of 96 decisions even with g = 0 — so the transcript is NOT audio-dependent; what the model
#     guarantees is that a wrong cross-attention changes it.)
#   * every residual-writing projection keeps its full scale (out_scale 1): self-attention, the other 31 cross-attentions and all MLPs
#     contribute at the level of a random-weight model
# sign(s_p) picks a_p or b_p; the margin is ~ g * |s_p| * (proj_a - proj_b) / rms(x).  (f, g, c) per architecture were calibrated
# with the reference CPU path (scripts/xplant_calibrate.py; profiles/r04_xplant_calibration.txt).
XPLANT = {"base.en": (80.0, 160.0, 1.0, 0.1), "large-v3-turbo": (64.0, 480.0, 1.0, 0.1), "large-v3": (80.0, 600.0, 1.0, 0.1)}      # (f, g, c, w); calibrated architectures only


def xplant_tables(te: np.ndarray, n_text_ctx: int, seed: int):
    """(u, a[p], b[p]) for a token-embedding matrix te [n_vocab, n] (f16)"""
    n = te.shape[1]
    u = np.random.default_rng(seed + 77).standard_normal(n).astype(np.float32)
    u /= np.linalg.norm(u)
    n_text = 50256                                  # ordinary text tokens only (the samplers suppress special tokens)
    proj = te[:n_text].astype(np.float32) @ u
    order = np.argsort(proj)
    return u, order[::-1][:n_text_ctx].copy(), order[:n_text_ctx].copy()


def read_token_embedding(f16_model: Path) -> np.ndarray:
    """decoder.token_embedding.weight [n_vocab, n] (f16) of an F16 model file written by write_f16_model"""
    with open(f16_model, "rb") as f:
        f.read(4 + 44)
        nm, nf = struct.unpack("ii", f.read(8))
        f.seek(nm * nf * 4, 1)
        (nv,) = struct.unpack("i", f.read(4))
        for _ in range(nv):
            (ln,) = struct.unpack("I", f.read(4))
            f.seek(ln, 1)
        while True:
            h = f.read(12)
            if len(h) < 12:
                raise RuntimeError("no token embedding in " + str(f16_model))
            nd, nl, ft = struct.unpack("iii", h)
            ne = struct.unpack("i" * nd, f.read(4 * nd))
            name = f.read(nl).decode()
            n = int(np.prod(ne)) * (2 if ft == 1 else 4)
            if name == "decoder.token_embedding.weight":
                return np.frombuffer(f.read(n), dtype=np.float16).reshape(ne[1], ne[0]).copy()
            f.seek(n, 1)


def xplant_candidates(arch: str, out_dir: Path | None = None, seed: int = 1234):
    """(a[p], b[p]): the two candidate tokens of every decoder position of the x-planted model of `arch` (made if necessary)"""
    f16 = make_model(arch, "f16", out_dir, seed, plant="x")
    _, a, b = xplant_tables(read_token_embedding(f16), ARCHS[arch][5], seed)
    return [int(t) for t in a], [int(t) for t in b]


def write_f16_model(path: Path, arch: str, seed: int = 1234, plant=False):
    """plant=True: a model with LARGE logit margins and a known transcript.  Random-weight models have near-Gaussian logits whose top-2
    margin falls below any backend's rounding differences every 10-20 steps, so free-running token sequences of two correct back ends
    part ways for reasons that have nothing to do with correctness.  Trained models have margins orders of
    magnitude larger; this plants that property: positional embedding p = 12 x the embedding of token planted_token(p), and the three
    residual-writing projections of every decoder layer scaled by 0.02, so that the final hidden state at position p points at
    planted_token(p) and the logits (token_embedding . LayerNorm(x), the same quantized matrix) put it ~50 above the runner-up.  Every
    kernel of the path still runs on full-size data; greedy AND beam search must then produce the planted sequence on any correct back end."""
    (n_vocab, n_audio_ctx, n_as, n_ah, n_al, n_text_ctx, n_ts, n_th, n_tl, n_mels) = ARCHS[arch]
    rng = _Rng(seed)
    with open(path, "wb") as f:
        f.write(struct.pack("I", 0x67676D6C))
        f.write(struct.pack("11i", n_vocab, n_audio_ctx, n_as, n_ah, n_al, n_text_ctx, n_ts, n_th, n_tl, n_mels, 1))
        # mel filterbank: any non-negative [n_mels x 201] matrix works for the CPU front end; triangular bands
        n_fft = 201
        filt = np.zeros((n_mels, n_fft), dtype=np.float32)
        edges = np.linspace(0, n_fft - 1, n_mels + 2)
        for m in range(n_mels):
            lo, c, hi = edges[m], edges[m + 1], edges[m + 2]
            k = np.arange(n_fft, dtype=np.float32)
            filt[m] = np.clip(np.minimum((k - lo) / max(c - lo, 1e-3), (hi - k) / max(hi - c, 1e-3)), 0, None) * (2.0 / (hi - lo))
        f.write(struct.pack("ii", n_mels, n_fft))
        f.write(filt.tobytes())
        # vocab: synthetic token strings; the loader appends the special tokens itself (whisper.cpp:1640-1672)
        n_file = 50257 if n_vocab >= 51865 else 50256
        f.write(struct.pack("i", n_file))
        for i in range(n_file):
            w = (" t%d" % i).encode()
            f.write(struct.pack("I", len(w)))
            f.write(w)

        def mat(name, out_f, in_f):          # 2-D weights: f16, N(0, 1/in)
            _w_tensor(f, name, rng.normal((out_f, in_f), 1.0 / np.sqrt(in_f)).astype(np.float16))

        def omat(name, out_f, in_f):         # the projections that write into the decoder's residual stream (scaled down in a planted model)
            _w_tensor(f, name, (rng.normal((out_f, in_f), 1.0 / np.sqrt(in_f)) * out_scale).astype(np.float16))

        out_scale = 1.0

        def vec(name, n, scale=0.02, base=0.0):  # biases / LN: f32
            _w_tensor(f, name, (rng.normal((n,), scale) + base).astype(np.float32))

        # encoder
        _w_tensor(f, "encoder.positional_embedding", rng.normal((n_audio_ctx, n_as), 0.02).astype(np.float32))
        _w_tensor(f, "encoder.conv1.weight", rng.normal((n_as, n_mels, 3), 1.0 / np.sqrt(3 * n_mels)).astype(np.float16))
        _w_tensor(f, "encoder.conv1.bias", rng.normal((n_as, 1), 0.02).astype(np.float32))
        _w_tensor(f, "encoder.conv2.weight", rng.normal((n_as, n_as, 3), 1.0 / np.sqrt(3 * n_as)).astype(np.float16))
        _w_tensor(f, "encoder.conv2.bias", rng.normal((n_as, 1), 0.02).astype(np.float32))
        vec("encoder.ln_post.weight", n_as, 0.02, 1.0)
        vec("encoder.ln_post.bias", n_as)
        for i in range(n_al):
            p = f"encoder.blocks.{i}."
            vec(p + "mlp_ln.weight", n_as, 0.02, 1.0); vec(p + "mlp_ln.bias", n_as)
            mat(p + "mlp.0.weight", 4 * n_as, n_as); vec(p + "mlp.0.bias", 4 * n_as)
            mat(p + "mlp.2.weight", n_as, 4 * n_as); vec(p + "mlp.2.bias", n_as)
            vec(p + "attn_ln.weight", n_as, 0.02, 1.0); vec(p + "attn_ln.bias", n_as)
            mat(p + "attn.query.weight", n_as, n_as); vec(p + "attn.query.bias", n_as)
            mat(p + "attn.key.weight", n_as, n_as)
            mat(p + "attn.value.weight", n_as, n_as); vec(p + "attn.value.bias", n_as)
            mat(p + "attn.out.weight", n_as, n_as); vec(p + "attn.out.bias", n_as)
        # decoder
        pe = rng.normal((n_text_ctx, n_ts), 0.02).astype(np.float32)
        # token embedding doubles as the logits matrix: larger scale so that logits are well separated
        te = rng.normal((n_vocab, n_ts), 0.05).astype(np.float16)
        xp = plant == "x"
        if xp:
            xf, xg, xc, xw = (float(v) for v in os.environ["XPLANT_PARAMS"].split(",")) if os.environ.get("XPLANT_PARAMS") else XPLANT[arch]      # (override: calibration runs)
            xu, xa, xb = xplant_tables(te, n_text_ctx, seed)
            pe = np.stack([xf * (te[xa[p]].astype(np.float32) + te[xb[p]].astype(np.float32)) for p in range(n_text_ctx)])
        elif plant:
            pe = np.stack([12.0 * te[planted_token(p)].astype(np.float32) for p in range(n_text_ctx)])
        _w_tensor(f, "decoder.positional_embedding", pe)
        _w_tensor(f, "decoder.token_embedding.weight", te)
        out_scale = 0.02 if (plant and not xp) else 1.0
        vec("decoder.ln.weight", n_ts, 0.02, 1.0)
        vec("decoder.ln.bias", n_ts)
        for i in range(n_tl):
            p = f"decoder.blocks.{i}."
            vec(p + "mlp_ln.weight", n_ts, 0.02, 1.0); vec(p + "mlp_ln.bias", n_ts)
            mat(p + "mlp.0.weight", 4 * n_ts, n_ts); vec(p + "mlp.0.bias", 4 * n_ts)
            omat(p + "mlp.2.weight", n_ts, 4 * n_ts); vec(p + "mlp.2.bias", n_ts, 0.02 * out_scale)
            vec(p + "attn_ln.weight", n_ts, 0.02, 1.0); vec(p + "attn_ln.bias", n_ts)
            mat(p + "attn.query.weight", n_ts, n_ts); vec(p + "attn.query.bias", n_ts)
            mat(p + "attn.key.weight", n_ts, n_ts)
            mat(p + "attn.value.weight", n_ts, n_ts); vec(p + "attn.value.bias", n_ts)
            omat(p + "attn.out.weight", n_ts, n_ts); vec(p + "attn.out.bias", n_ts, 0.02 * out_scale)
            vec(p + "cross_attn_ln.weight", n_ts, 0.02, 1.0); vec(p + "cross_attn_ln.bias", n_ts)
            if xp and i == n_tl - 1:
                # the deciding block (see XPLANT above)
                _w_tensor(f, p + "cross_attn.query.weight", (rng.normal((n_ts, n_ts), 1.0 / np.sqrt(n_ts)) * xc).astype(np.float16)); vec(p + "cross_attn.query.bias", n_ts)
                _w_tensor(f, p + "cross_attn.key.weight", (rng.normal((n_ts, n_ts), 1.0 / np.sqrt(n_ts)) * xc).astype(np.float16))
                wv = rng.normal((n_ts, n_ts), 1.0 / np.sqrt(n_ts))
                wv[0, :] *= xw                               # row 0 = w of the comment above, small against the bias: s_p = 1 + O(w)
                _w_tensor(f, p + "cross_attn.value.weight", wv.astype(np.float16))
                bv = rng.normal((n_ts,), 0.02).astype(np.float32)
                bv[0] = 1.0                                  # s_p = sum_k softmax_k (w . enc_k) + 1: a stable sign, still every step of the attention arithmetic
                _w_tensor(f, p + "cross_attn.value.bias", bv)
                wo = rng.normal((n_ts, n_ts), 1.0 / np.sqrt(n_ts))
                wo[:, 0] = xg * xu
                _w_tensor(f, p + "cross_attn.out.weight", wo.astype(np.float16)); vec(p + "cross_attn.out.bias", n_ts, 0.02)
                continue
            mat(p + "cross_attn.query.weight", n_ts, n_ts); vec(p + "cross_attn.query.bias", n_ts)
            mat(p + "cross_attn.key.weight", n_ts, n_ts)
            mat(p + "cross_attn.value.weight", n_ts, n_ts); vec(p + "cross_attn.value.bias", n_ts)
            omat(p + "cross_attn.out.weight", n_ts, n_ts); vec(p + "cross_attn.out.bias", n_ts, 0.02 * out_scale)


def make_model(arch: str, qtype: str, out_dir: Path | None = None, seed: int = 1234, quantize_bin: Path | None = None, plant=False) -> Path:
    """Create (or reuse) <out_dir>/synth-<arch>[-planted]-<qtype>.bin and return its path."""
    assert arch in ARCHS and qtype in QTYPES
    out_dir = Path(out_dir or os.environ.get("WHISPER_SYNTH_DIR", "/tmp/whisper_synth"))
    out_dir.mkdir(parents=True, exist_ok=True)
    tag = f"{arch}-xplanted" if plant == "x" else (f"{arch}-planted" if plant else arch)
    f16 = out_dir / f"synth-{tag}-f16.bin"
    if not f16.exists():
        tmp = f16.with_suffix(".tmp")
        write_f16_model(tmp, arch, seed, plant)
        tmp.rename(f16)
    if qtype == "f16":
        return f16
    q = out_dir / f"synth-{tag}-{qtype}.bin"
    if not q.exists():
        # the reference application's own quantizer, installed next to the unmodified libwhisper the plugin drops into
        qb = Path(quantize_bin or ROOT / "whisper.cpp_amd" / "host" / "_whisper" / "whisper-quantize")
        tmp = q.with_suffix(".tmp")
        r = subprocess.run([str(qb), str(f16), str(tmp), qtype], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 or not tmp.exists():
            raise RuntimeError(f"whisper-quantize failed:\n{r.stdout[-2000:]}")
        tmp.rename(q)
    return q


def synth_audio(n_samples: int = 16000 * 11, seed: int = 7) -> np.ndarray:
    """Deterministic speech-like test signal (chirps + amplitude modulation + noise), f32 in [-1, 1], 16 kHz."""
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    rng = np.random.default_rng(seed)
    x = 0.35 * np.sin(2 * np.pi * (180 + 60 * np.sin(2 * np.pi * 0.7 * t)) * t)
    x += 0.20 * np.sin(2 * np.pi * (700 + 300 * np.sin(2 * np.pi * 1.3 * t)) * t)
    x += 0.10 * np.sin(2 * np.pi * (2300 + 500 * np.sin(2 * np.pi * 0.4 * t)) * t)
    x *= 0.5 * (1 + np.sin(2 * np.pi * 3.1 * t)) * (t % 2.0 < 1.6)
    x += 0.02 * rng.standard_normal(n_samples)
    return np.clip(x, -1, 1).astype(np.float32)


def write_wav(path: Path, pcm: np.ndarray):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.clip(pcm, -1, 1) * 32767).astype("<i2").tobytes())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="base.en", choices=sorted(ARCHS))
    ap.add_argument("--qtype", default="q5_0", choices=QTYPES)
    ap.add_argument("--out-dir", default=None)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--plant", action="store_true", help="large-margin model with a known transcript (see write_f16_model)")
    ap.add_argument("--xplant", action="store_true", help="model whose token decisions are carried by cross-attention (see XPLANT)")
    a = ap.parse_args()
    print(make_model(a.arch, a.qtype, a.out_dir, a.seed, plant="x" if a.xplant else a.plant))
