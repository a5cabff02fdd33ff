"""Generate the golden vectors that pin oracle/oracle.c, FROM THE REFERENCE ITSELF (oracle/_ref, built by
oracle/Makefile from /root/reference).  Run in the build container:  python tests/golden/make_golden.py

  tests/golden/ops.npz     op-level cases computed by the reference CPU backend through ggml graphs
                           (tests/native/bin/op_parity with OP_PARITY_DUMP; inputs are raw ggml block bytes)
  tests/golden/blocks.npz  block-level cases from the reference's exported functions (ctypes on libggml-base /
                           libggml-cpu): dequantize_row_*, quantize_row_*_ref, the AVX2 quantize_row_q8_0,
                           quantize_row_q8_K, ggml_vec_dot_*, ggml_fp32_to_fp16_row, ggml_table_gelu_f16
"""
import ctypes as C
import json
import os
import subprocess
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = ROOT / "oracle" / "_ref"


def ops_npz():
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, OP_PARITY_DUMP=d, GGML_MI355X_PLUGIN="cpu")
        subprocess.run([str(ROOT / "tests/native/bin/op_parity")], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        manifest = []
        for line in open(Path(d) / "manifest.jsonl"):
            m = json.loads(line)
            manifest.append(m)
            for i, leaf in enumerate(m["leaves"]):
                out[f"{m['case']}.leaf{i}"] = np.fromfile(Path(d) / leaf["file"], dtype=np.uint8)
            for i, o in enumerate(m["outs"]):
                out[f"{m['case']}.out{i}"] = np.fromfile(Path(d) / o["file"], dtype=np.float32)
        out["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
        np.savez_compressed(HERE / "ops.npz", **out)
        print("ops.npz:", len(manifest), "cases")


def blocks_npz():
    base = C.CDLL(str(REF / "libggml-base.so"), mode=C.RTLD_GLOBAL)
    cpu = C.CDLL(str(REF / "libggml-cpu.so"), mode=C.RTLD_GLOBAL)
    cpu.ggml_cpu_init()
    rng = np.random.default_rng(20260921)
    n = 1024
    x = (rng.standard_normal(n) * np.repeat(rng.uniform(0.01, 30, n // 32), 32)).astype(np.float32)
    x[64:96] = 0.0                                     # an all-zero block
    x[96] = 127.5; x[97] = -0.5; x[98] = 2.5           # exact .5 ties after scaling (amax = 127.5 -> id = 127/127.5)
    w = (rng.standard_normal(n) * 0.05).astype(np.float32)
    out = {"x": x, "w": w}
    sizes = {"q4_0": 18 * n // 32, "q5_0": 22 * n // 32, "q8_0": 34 * n // 32, "q4_K": 144 * n // 256}
    for t, nb in sizes.items():
        blk = np.zeros(nb, dtype=np.uint8)
        getattr(base, f"quantize_row_{t}_ref")(w.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p), C.c_int64(n))
        deq = np.zeros(n, dtype=np.float32)
        getattr(base, f"dequantize_row_{t}")(blk.ctypes.data_as(C.c_void_p), deq.ctypes.data_as(C.c_void_p), C.c_int64(n))
        out[f"wblk_{t}"] = blk
        out[f"wdeq_{t}"] = deq
    # a wider Q4_K weight-quantizer case (make_qkx2_quants search): varied magnitudes per 32-block, an all-zero super-block,
    # an all-positive 32-block (min clamps to 0), a constant block, a block with one outlier
    xk = (rng.standard_normal(16 * 256) * np.repeat(rng.uniform(0.001, 3.0, 16 * 8), 32)).astype(np.float32)
    xk[256:512] = 0.0
    xk[512:544] = np.abs(xk[512:544]) + 0.25
    xk[544:576] = 0.731
    xk[576:608] *= 0.01; xk[590] = 4.0
    kb = np.zeros(144 * 16, dtype=np.uint8)
    base.quantize_row_q4_K_ref(xk.ctypes.data_as(C.c_void_p), kb.ctypes.data_as(C.c_void_p), C.c_int64(len(xk)))
    out["q4k_x"] = xk
    out["q4k_blk"] = kb
    a8 = np.zeros(34 * n // 32, dtype=np.uint8)
    cpu.quantize_row_q8_0(x.ctypes.data_as(C.c_void_p), a8.ctypes.data_as(C.c_void_p), C.c_int64(n))      # AVX2 path
    out["act_q8_0"] = a8
    aK = np.zeros(292 * n // 256, dtype=np.uint8)
    cpu.quantize_row_q8_K(x.ctypes.data_as(C.c_void_p), aK.ctypes.data_as(C.c_void_p), C.c_int64(n))
    out["act_q8_K"] = aK
    for t in sizes:
        s = C.c_float(0)
        act = aK if t == "q4_K" else a8
        fn = getattr(cpu, f"ggml_vec_dot_{t}_q8_{'K' if t == 'q4_K' else '0'}")
        fn(C.c_int(n), C.byref(s), C.c_size_t(0), out[f"wblk_{t}"].ctypes.data_as(C.c_void_p), C.c_size_t(0),
           act.ctypes.data_as(C.c_void_p), C.c_size_t(0), C.c_int(1))
        out[f"dot_{t}"] = np.float32(s.value)
    h = np.zeros(n, dtype=np.uint16)
    base.ggml_fp32_to_fp16_row(x.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), C.c_int64(n))
    out["x_f16"] = h
    tab = (C.c_uint16 * 65536).in_dll(cpu, "ggml_table_gelu_f16")
    out["gelu_table"] = np.frombuffer(tab, dtype=np.uint16).copy()
    np.savez_compressed(HERE / "blocks.npz", **out)
    print("blocks.npz:", sorted(out))


def mel_npz():
    """the reference's own log-mel front end on real speech: samples/jfk.wav (11 s, 16 kHz mono, the file the north star names) with
    the real 80-band filterbank of models/for-tests-ggml-base.en.bin, through oracle/_ref/mel_ref (which includes the reference's
    src/whisper.cpp in place).  Stored: the PCM (int16), the filterbank, the mel up to the last frame that sees audio + its tail value."""
    import struct
    import wave
    ref = Path(os.environ.get("WHISPER_REF", "/root/reference"))
    w = wave.open(str(ref / "samples" / "jfk.wav"))
    assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
    pcm16 = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    model = ref / "models" / "for-tests-ggml-base.en.bin"
    with open(model, "rb") as f:
        f.read(4 + 44)
        nm, nf = struct.unpack("ii", f.read(8))
        filt = np.frombuffer(f.read(nm * nf * 4), dtype=np.float32).reshape(nm, nf).copy()
    with tempfile.TemporaryDirectory() as d:
        (pcm16.astype(np.float32) / 32768.0).tofile(Path(d) / "pcm.f32")          # the conversion whisper-cli applies (examples/common-whisper.cpp)
        subprocess.run([str(REF / "mel_ref"), str(model), str(Path(d) / "pcm.f32"), str(Path(d) / "mel.bin"), "4"], check=True, stderr=subprocess.DEVNULL)
        raw = (Path(d) / "mel.bin").read_bytes()
    n_mel, n_len, n_org = struct.unpack("iii", raw[:12])
    mel = np.frombuffer(raw[12:], dtype=np.float32).reshape(n_mel, n_len)
    keep = 1104
    assert np.all(mel[:, keep:] == mel[0, -1])
    np.savez_compressed(HERE / "mel.npz", pcm16=pcm16, filters=filt, mel_head=mel[:, :keep].copy(), tail_value=np.float32(mel[0, -1]),
                        n_len=np.int32(n_len), n_len_org=np.int32(n_org))
    print("mel.npz:", n_mel, n_len, n_org, "tail", mel[0, -1])


def vad_npz():
    """the reference's only real-weight known-answer test on this path (tests/test-vad.cpp:31,39): the silero VAD graph (src/whisper.cpp:
    4545-4680) with its TRAINED weights (models/for-tests-silero-v6.2.0-ggml.bin, 885 KB — a data fixture, stored byte for byte because
    the GPU box has no /root/reference) on samples/jfk.wav: the 344 probabilities and 4 speech segments of the reference CPU path, produced
    by tests/native/bin/vad_parity (which includes the reference's src/whisper.cpp in place) in its CPU-only mode."""
    import json
    ref = Path(os.environ.get("WHISPER_REF", "/root/reference"))
    model = ref / "models" / "for-tests-silero-v6.2.0-ggml.bin"
    pcm16 = np.load(HERE / "mel.npz")["pcm16"]
    with tempfile.TemporaryDirectory() as d:
        (pcm16.astype(np.float32) / 32768.0).tofile(Path(d) / "pcm.f32")
        r = subprocess.run([str(HERE.parent / "native" / "bin" / "vad_parity"), str(model), str(Path(d) / "pcm.f32"), "cpu"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    j = json.loads(r.stdout)["cpu"]
    assert j["n_probs"] == 344 and len(j["segments"]) == 4, (j["n_probs"], j["segments"])          # the reference test's own assertions
    np.savez_compressed(HERE / "vad.npz", model=np.frombuffer(model.read_bytes(), dtype=np.uint8), probs=np.array(j["probs"], dtype=np.float32),
                        segments=np.array(j["segments"], dtype=np.float32))
    print("vad.npz:", j["n_probs"], "probabilities,", j["segments"])


if __name__ == "__main__":
    ops_npz()
    blocks_npz()
    mel_npz()
    vad_npz()
